"""ctypes binding of the C ABI in include/ace355.h (csrc/libace355.so).

There is no CPU or PyTorch fallback here: if the library is missing or a call fails, a
``RuntimeError`` carrying ``ace355_last_error()`` is raised (the reference's seam turns backend
exceptions into its PyTorch fallback, handler/service_generate_execute.py:189-191).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_LIB: Optional[C.CDLL] = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libace355.so")

OK = 0
DTYPE_F32, DTYPE_BF16 = 0, 1
MAX_BLOCKS = 8
MAX_SLOTS = 32


class DitConfigC(C.Structure):
    _fields_ = [
        ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32), ("num_layers", C.c_int32),
        ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32), ("head_dim", C.c_int32),
        ("sliding_window", C.c_int32), ("patch_size", C.c_int32), ("in_channels", C.c_int32),
        ("out_channels", C.c_int32), ("rms_norm_eps", C.c_float), ("rope_theta", C.c_float),
        ("sliding_layer_mask", C.c_uint64),
    ]


class SampleParamsC(C.Structure):
    _fields_ = [
        ("num_steps", C.c_int32), ("t_sched_host", C.POINTER(C.c_float)), ("guidance_scale", C.c_float),
        ("cfg_interval_start", C.c_float), ("cfg_interval_end", C.c_float), ("infer_method", C.c_int32),
        ("use_adg", C.c_int32), ("cond_slot", C.c_int32), ("null_slot", C.c_int32),
        ("cover_switch_step", C.c_int32), ("non_cover_slot", C.c_int32), ("ctx_non_cover_dev", C.c_void_p),
        ("sde_noise_dev", C.c_void_p), ("cond_slots_host", C.POINTER(C.c_int32)), ("non_cover_slots_host", C.POINTER(C.c_int32)),
        ("sde_next_from_sched", C.c_int32),
    ]


class CondConfigC(C.Structure):
    _fields_ = [
        ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32), ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32),
        ("head_dim", C.c_int32), ("text_hidden_dim", C.c_int32), ("timbre_hidden_dim", C.c_int32),
        ("num_lyric_layers", C.c_int32), ("num_timbre_layers", C.c_int32), ("sliding_window", C.c_int32),
        ("sliding_layer_mask", C.c_uint64), ("rms_norm_eps", C.c_float), ("rope_theta", C.c_float),
    ]


class DetokConfigC(C.Structure):
    _fields_ = [
        ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32), ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32),
        ("head_dim", C.c_int32), ("num_layers", C.c_int32), ("pool_window_size", C.c_int32), ("out_dim", C.c_int32),
        ("sliding_window", C.c_int32), ("sliding_layer_mask", C.c_uint64), ("rms_norm_eps", C.c_float), ("rope_theta", C.c_float),
    ]


class VaeConfigC(C.Structure):
    _fields_ = [
        ("decoder_channels", C.c_int32), ("decoder_input_channels", C.c_int32), ("audio_channels", C.c_int32),
        ("num_blocks", C.c_int32), ("channel_multiples", C.c_int32 * MAX_BLOCKS),
        ("upsampling_ratios", C.c_int32 * MAX_BLOCKS),
    ]


# name -> (restype, argtypes); every symbol include/ace355.h declares
SIGNATURES = {
    "ace355_last_error": (C.c_char_p, []),
    "ace355_version": (C.c_int, []),
    "ace355_dit_create": (C.c_int, [C.POINTER(DitConfigC), C.POINTER(C.c_void_p)]),
    "ace355_dit_destroy": (None, [C.c_void_p]),
    "ace355_dit_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int64, C.c_int]),
    "ace355_dit_finalize": (C.c_int, [C.c_void_p]),
    "ace355_dit_set_condition": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ace355_dit_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ace355_dit_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(SampleParamsC),
                                    C.c_void_p, C.POINTER(C.c_float), C.c_void_p]),
    "ace355_dit_set_tap": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ace355_dit_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "ace355_dit_set_graph": (C.c_int, [C.c_void_p, C.c_int]),
    "ace355_dit_set_norm_fold": (C.c_int, [C.c_void_p, C.c_int]),
    "ace355_box_probe_mfma": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "ace355_gemm_set_k_rotation": (C.c_int, [C.c_int]),
    "ace355_dit_set_dual": (C.c_int, [C.c_void_p, C.c_int]),
    "ace355_dit_dual_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ace355_dit_set_dedup": (C.c_int, [C.c_void_p, C.c_int]),
    "ace355_dit_dedup_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ace355_dit_graph_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ace355_dit_poll_errors": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ace355_dit_trim_slots": (C.c_int, [C.c_void_p, C.c_int]),
    "ace355_dit_set_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "ace355_dit_get_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ace355_vae_create": (C.c_int, [C.POINTER(VaeConfigC), C.POINTER(C.c_void_p)]),
    "ace355_vae_destroy": (None, [C.c_void_p]),
    "ace355_vae_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int64, C.c_int]),
    "ace355_vae_finalize": (C.c_int, [C.c_void_p]),
    "ace355_vae_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ace355_vae_hop": (C.c_int, [C.c_void_p]),
    "ace355_vae_set_decode_budget": (C.c_int, [C.c_void_p, C.c_int64, C.c_int]),
    "ace355_vae_decode_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int64)]),
    "ace355_vae_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "ace355_vae_latent_frames": (C.c_int, [C.c_void_p, C.c_int64]),
    "ace355_vae_set_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "ace355_vae_get_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ace355_cond_create": (C.c_int, [C.POINTER(CondConfigC), C.POINTER(C.c_void_p)]),
    "ace355_cond_destroy": (None, [C.c_void_p]),
    "ace355_cond_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int64, C.c_int]),
    "ace355_cond_finalize": (C.c_int, [C.c_void_p]),
    "ace355_cond_out_len": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int]),
    "ace355_cond_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_int,
                                     C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int32),
                                     C.c_void_p]),
    "ace355_tok_create": (C.c_int, [C.POINTER(DetokConfigC), C.POINTER(C.c_void_p)]),
    "ace355_tok_destroy": (None, [C.c_void_p]),
    "ace355_tok_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int64, C.c_int]),
    "ace355_tok_finalize": (C.c_int, [C.c_void_p]),
    "ace355_tok_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ace355_detok_create": (C.c_int, [C.POINTER(DetokConfigC), C.POINTER(C.c_void_p)]),
    "ace355_detok_destroy": (None, [C.c_void_p]),
    "ace355_detok_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int64, C.c_int]),
    "ace355_detok_finalize": (C.c_int, [C.c_void_p]),
    "ace355_detok_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ace355_peak_normalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "ace355_latent_check": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.c_void_p]),
    "ace355_normalize_audio": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "ace355_audio_interleave": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "ace355_flac_bound": (C.c_int64, [C.c_int64, C.c_int]),
    "ace355_flac_encode_pcm16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "ace355_flac_info": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ace355_flac_decode_pcm16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int]),
    "ace355_wav_bound": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "ace355_wav_encode": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "ace355_save_audio_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_void_p]),
    "ace355_gemm_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ace355_gemm_bf16_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ace355_gemm_bf16_residual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "ace355_gemm_bf16_headnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "ace355_mx_quantize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ace355_mx_rows_pad": (C.c_int, [C.c_int]),
    "ace355_gemm_mxfp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_void_p]),
    "ace355_rmsnorm_mod": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ace355_headnorm_rope": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int,
                                       C.c_float, C.c_void_p]),
    "ace355_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_float, C.c_void_p]),
    "ace355_attention_masked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_float, C.POINTER(C.c_int32), C.c_void_p]),
    "ace355_linear_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "ace355_apg_euler_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                        C.c_int, C.c_void_p]),
    "ace355_conv1d_nlc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}


def lib() -> C.CDLL:
    """Load csrc/libace355.so (once).  Raises RuntimeError if it has not been built."""
    global _LIB
    if _LIB is None:
        # torch bundles its own libamdhip64.so.7; device pointers and streams are shared with torch, so both must
        # run on ONE HIP runtime instance: load torch's first (same SONAME -> our NEEDED entry resolves to it).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"ace355 native library not found at {LIB_PATH}. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the header and the library drift apart
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def last_error() -> str:
    msg = lib().ace355_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str = "") -> None:
    if rc != OK:
        raise RuntimeError(f"ace355 native call failed ({what}, code {rc}): {last_error()}")


def gemm_set_k_rotation(mode: int) -> int:
    """`ace355_gemm_set_k_rotation` (include/ace355.h): 0 off, 1 launches with N <= 2048 (default), 2 every one-round launch; returns the
    previous mode.  Process-wide."""
    return int(lib().ace355_gemm_set_k_rotation(int(mode)))


def current_stream_ptr() -> int:
    """The caller's current HIP stream as an integer handle (use the caller's stream)."""
    import torch
    return int(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> int:
    """Device/host address of a contiguous torch tensor (0 for None)."""
    if t is None:
        return 0
    assert t.is_contiguous(), "ace355: tensors crossing the C ABI must be contiguous"
    return int(t.data_ptr())
