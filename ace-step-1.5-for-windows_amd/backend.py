"""The reference's backend seam, re-implemented for the native MI355X path.

The reference selects an alternate DiT / VAE backend through handler flags and a pair of mixin methods introduced
for MLX:
  * ``_init_mlx_dit`` / ``_init_mlx_vae``   (handler/mlx_dit_init.py:9-43, handler/mlx_vae_init.py:12-96)
  * ``_mlx_run_diffusion``                  (handler/diffusion.py:18-140), called from
    ``_execute_service_generate_diffusion`` (handler/service_generate_execute.py:144-194)
  * ``_mlx_vae_decode`` via ``tiled_decode`` (handler/vae_decode.py:39-48, handler/generate_music_decode.py:165-167)

``NativeDitMixin`` / ``NativeVaeMixin`` provide the same-shaped methods over the C ABI, so a maintainer mixes them
into ``AceStepHandler``.  Because the reference's Python cannot travel to the GPU box,
``NativeHandler`` is this repo's own counterpart of the handler for that path: ``initialize_service`` /
``service_generate`` / ``generate_music`` with the reference's keyword names and return-dict shapes, fed by
pre-embedded conditioning tensors instead of raw text (the text/LM stages are out of scope).

Error contract (reference): backend failure = Python exception, caught at the seam, fall back / error payload;
``generate_music`` never raises (handler/generate_music.py:181-190).  The native methods themselves never fall back
to a CPU implementation: a missing library or a failed HIP call raises ``RuntimeError``.
"""
from __future__ import annotations

import contextlib
import inspect
import logging
import time
import traceback
from typing import Any, Dict, List, Optional, Sequence, Union

import torch

from .config import DitConfig, VaeConfig

logger = logging.getLogger("ace355")

MAX_BATCH_SIZE = 8  # handler/service_generate_request.py:12


class NativeDitMixin:
    """Required host attributes: ``model`` (with ``.decoder``, ``.null_condition_emb``, ``.config``), ``device``, ``dtype``."""

    use_native_dit: bool = False
    native_dit = None

    def _init_native_dit(self) -> bool:
        """Counterpart of ``_init_mlx_dit``: convert weights from the already-loaded PyTorch module. Never raises."""
        try:
            quant = getattr(self, "quantization", None)
            # the reference's quantization knob (init_service_loader.py:89-113): "fp8_weight_only" has a native counterpart (same
            # numerics: per-channel e4m3 weights, bf16 arithmetic); the int8 schemes do not
            if getattr(self, "use_lora", False) or (quant and quant != "fp8_weight_only") or getattr(self, "offload_to_cpu", False):
                logger.info("[native-dit] LoRA / int8 quantization / CPU offload active; keeping the PyTorch path")
                self.use_native_dit, self.native_dit = False, None
                return False
            from .dit import NativeDit
            cfg = DitConfig.from_reference(self.model.config) if not isinstance(getattr(self.model, "config", None), DitConfig) \
                else self.model.config
            dit = NativeDit(cfg, self.device)
            dit.load_state_dict(self.model.decoder.state_dict())
            if quant == "fp8_weight_only":
                dit.set_precision("fp8_weight_only")
            self.native_dit, self.use_native_dit = dit, True
            logger.info("[native-dit] decoder packed for gfx950 (%d layers, hidden %d)", cfg.num_hidden_layers, cfg.hidden_size)
            return True
        except Exception as exc:  # same policy as handler/mlx_dit_init.py:36-43
            logger.warning("[native-dit] init failed (%s: %s); PyTorch path stays active", type(exc).__name__, exc)
            self.use_native_dit, self.native_dit = False, None
            return False

    def _model_honours_timesteps(self) -> bool:
        """Whether the loaded checkpoint's own ``generate_audio`` reads an explicit ``timesteps=``: the sft and turbo models do
        (sft/modeling_acestep_v15_base.py:1864-1875, turbo :1807-1865); the BASE model has no such parameter - it lands in
        ``**kwargs`` and the schedule is always linspace + shift (base/modeling_acestep_v15_base.py:1812, 1864-1867).  The
        native path must not diverge from the checkpoint it stands in for.  ``model_variant`` ("base" | "sft" | "turbo") on
        the host overrides the signature probe (``NativeHandler`` has no reference module to probe)."""
        variant = getattr(self, "model_variant", None)
        if variant is not None:
            return variant in ("sft", "turbo")
        ga = getattr(getattr(self, "model", None), "generate_audio", None)
        if ga is None:
            return True
        try:
            return "timesteps" in inspect.signature(ga).parameters
        except (TypeError, ValueError):
            return True

    def _is_turbo(self) -> bool:
        """handler/init_service_catalog.py:69-73 (`is_turbo_model`): a ``config.is_turbo`` flag on the host, or
        ``model_variant == "turbo"``."""
        return bool(getattr(getattr(self, "config", None), "is_turbo", False)) or getattr(self, "model_variant", None) == "turbo"

    def _native_run_diffusion(
        self,
        encoder_hidden_states,
        encoder_attention_mask,
        context_latents,
        src_latents,
        seed,
        infer_method: str = "ode",
        shift: float = 3.0,
        timesteps=None,
        audio_cover_strength: float = 1.0,
        encoder_hidden_states_non_cover=None,
        encoder_attention_mask_non_cover=None,
        context_latents_non_cover=None,
        disable_tqdm: bool = False,
        # base/sft knobs the MLX path lacks (handler/service_generate_execute.py:78-104)
        infer_steps: int = 30,
        guidance_scale: float = 7.0,
        cfg_interval_start: float = 0.0,
        cfg_interval_end: float = 1.0,
        use_adg: bool = False,
        cover_noise_strength: float = 0.0,
    ) -> Dict[str, Any]:
        """Same contract as ``_mlx_run_diffusion`` (handler/diffusion.py:18-140): returns
        ``{"target_latents": Tensor[B,T,64] on self.device in self.dtype, "time_costs": {...}}``.
        The attention masks are accepted and unused: the DiT discards them (modeling_acestep_v15_base.py:1384-1385)."""
        _ = encoder_attention_mask, encoder_attention_mask_non_cover, disable_tqdm
        for attr in ("native_dit", "device", "dtype", "model"):
            if not hasattr(self, attr):
                raise AttributeError(f"NativeDitMixin host is missing required attribute '{attr}'")
        if self.native_dit is None:
            raise RuntimeError("native DiT backend is not initialised (call _init_native_dit)")
        if infer_method not in {"ode", "sde"}:
            raise ValueError(f"Unsupported infer_method '{infer_method}'. Expected 'ode' or 'sde'.")
        if timesteps is not None and not (hasattr(timesteps, "__iter__") or hasattr(timesteps, "tolist")):
            raise TypeError("timesteps must be iterable, tensor-like, or None")
        if encoder_hidden_states.shape[0] != context_latents.shape[0]:
            raise ValueError("Batch dimension mismatch: encoder_hidden_states and context_latents must share dim 0")
        if encoder_hidden_states.shape[0] != src_latents.shape[0]:
            raise ValueError("Batch dimension mismatch: encoder_hidden_states and src_latents must share dim 0")
        if encoder_hidden_states_non_cover is not None and encoder_hidden_states_non_cover.shape[0] != encoder_hidden_states.shape[0]:
            raise ValueError("Batch dimension mismatch: encoder_hidden_states_non_cover must share dim 0 with encoder_hidden_states")
        if context_latents_non_cover is not None and context_latents_non_cover.shape[0] != context_latents.shape[0]:
            raise ValueError("Batch dimension mismatch: context_latents_non_cover must share dim 0 with context_latents")
        ts_list = None
        if timesteps is not None:
            ts_list = timesteps.tolist() if hasattr(timesteps, "tolist") else list(timesteps)
        if ts_list is not None and not self._is_turbo() and not self._model_honours_timesteps():
            logger.info("[native-dit] base checkpoint: explicit timesteps are ignored, as its generate_audio does")
            ts_list = None
        if self._is_turbo():
            # turbo checkpoints (handler init_service_catalog.py:69-73 `is_turbo_model`): the model's own generate_audio has no
            # CFG / step count, only the shift -> 8-step table (turbo modeling file :1780-1995); same native loop
            from .dit import generate_latents_turbo
            out = generate_latents_turbo(
                self.native_dit, encoder_hidden_states, context_latents, seed=seed, shift=shift, timesteps=ts_list,
                infer_method=infer_method, audio_cover_strength=audio_cover_strength, cover_noise_strength=cover_noise_strength,
                src_latents=src_latents, encoder_hidden_states_non_cover=encoder_hidden_states_non_cover,
                context_latents_non_cover=context_latents_non_cover)
            out["target_latents"] = out["target_latents"].to(device=self.device, dtype=self.dtype)
            return out
        from .dit import generate_latents
        out = generate_latents(
            self.native_dit, self.model.null_condition_emb.detach(), encoder_hidden_states, context_latents, seed=seed,
            infer_method=infer_method, infer_steps=infer_steps, diffusion_guidance_sale=guidance_scale,
            cfg_interval_start=cfg_interval_start, cfg_interval_end=cfg_interval_end, use_adg=use_adg, shift=shift,
            timesteps=ts_list, audio_cover_strength=audio_cover_strength, cover_noise_strength=cover_noise_strength,
            src_latents=src_latents, encoder_hidden_states_non_cover=encoder_hidden_states_non_cover,
            context_latents_non_cover=context_latents_non_cover)
        out["target_latents"] = out["target_latents"].to(device=self.device, dtype=self.dtype)
        return out


class NativeVaeMixin:
    """Required host attributes: ``vae`` (state_dict + config) or ``vae_state_dict``/``vae_config``, ``device``."""

    use_native_vae: bool = False
    native_vae = None

    def _init_native_vae(self) -> bool:
        """Counterpart of ``_init_mlx_vae``: fuse weight-norm + pack from the loaded ``AutoencoderOobleck``. Never raises."""
        try:
            from .vae import NativeVae
            vae_mod = getattr(self, "vae", None)
            cfg = getattr(self, "vae_config", None)
            if cfg is None:
                cfg = VaeConfig.from_reference(vae_mod.config)
            sd = getattr(self, "vae_state_dict", None) or vae_mod.state_dict()
            nv = NativeVae(cfg, self.device)
            nv.load_state_dict(sd)
            self.native_vae, self.use_native_vae = nv, True
            logger.info("[native-vae] decoder packed for gfx950 (hop %d)", nv.hop)
            return True
        except Exception as exc:
            logger.warning("[native-vae] init failed (%s: %s); PyTorch path stays active", type(exc).__name__, exc)
            self.use_native_vae, self.native_vae = False, None
            return False

    def _native_vae_decode(self, latents: torch.Tensor) -> torch.Tensor:
        """``latents [B,64,T]`` -> fp32 waveform ``[B,2,hop*T]`` on the device (``_mlx_vae_decode`` returns CPU fp32;
        both are accepted downstream: ``.float()``, ``amax``, ``.cpu()`` follow, handler/generate_music_decode.py:191)."""
        if self.native_vae is None:
            raise RuntimeError("native VAE backend is not initialised (call _init_native_vae)")
        return self.native_vae.decode(latents)

    def _native_vae_encode_sample(self, audio: torch.Tensor) -> torch.Tensor:
        """``audio [B,2,L]`` -> sampled latents fp32 ``[B,64,T]`` (counterpart of ``_mlx_vae_encode_sample``,
        handler/vae_encode.py:33: whole-sequence encode; the 30 s / 2 s-overlap tiling is a memory workaround)."""
        if self.native_vae is None or not getattr(self.native_vae, "has_encoder", False):
            raise RuntimeError("native VAE encoder is not initialised (encoder.* weights missing)")
        return self.native_vae.encode(audio)

    def tiled_encode(self, audio, chunk_size=None, overlap=None, offload_latent_to_cpu=True):
        """Same signature as handler/vae_encode.py:15-86.  Native fast path first; on failure the host's PyTorch
        implementation (``VaeEncodeMixin.tiled_encode`` further up the MRO) runs, exactly like the MLX seam (:28-45)."""
        if self.use_native_vae and self.native_vae is not None and getattr(self.native_vae, "has_encoder", False):
            input_was_2d = audio.dim() == 2
            try:
                result = self._native_vae_encode_sample(audio.unsqueeze(0) if input_was_2d else audio)
                return result.squeeze(0) if input_was_2d else result
            except Exception as exc:
                logger.warning("[tiled_encode] native VAE encode failed (%s: %s)", type(exc).__name__, exc)
        parent = getattr(super(), "tiled_encode", None)
        if parent is None:
            raise RuntimeError("no VAE encode backend available")
        return parent(audio, chunk_size=chunk_size, overlap=overlap, offload_latent_to_cpu=offload_latent_to_cpu)

    def tiled_decode(self, latents, chunk_size: Optional[int] = None, overlap: int = 64, offload_wav_to_cpu: Optional[bool] = None):
        """Same signature as handler/vae_decode.py:16-85.  Native fast path first (whole-sequence decode, which equals
        the overlap-discard tiling up to fp summation order); on failure the host's PyTorch
        ``_tiled_decode_inner`` runs if the host has one, exactly like the MLX seam (handler/vae_decode.py:39-48)."""
        if self.use_native_vae and self.native_vae is not None:
            try:
                return self._native_vae_decode(latents)
            except Exception as exc:
                logger.warning("[tiled_decode] native VAE decode failed (%s: %s)", type(exc).__name__, exc)
                if not hasattr(self, "_tiled_decode_inner"):
                    raise
        if not hasattr(self, "_tiled_decode_inner"):
            raise RuntimeError("no VAE decode backend available")
        if chunk_size is None:
            chunk_size = self._get_auto_decode_chunk_size()
        if offload_wav_to_cpu is None:
            offload_wav_to_cpu = self._should_offload_wav_to_cpu()
        return self._tiled_decode_inner(latents, chunk_size, overlap, offload_wav_to_cpu)


class _ModelShell:
    """What the mixins read from ``self.model``: decoder state dict, null embedding, config."""

    class _Dec:
        def __init__(self, sd):
            self._sd = sd

        def state_dict(self):
            return self._sd

    def __init__(self, cfg: DitConfig, decoder_sd: Dict[str, torch.Tensor], null_condition_emb: torch.Tensor):
        self.config = cfg
        self.decoder = self._Dec(decoder_sd)
        self.null_condition_emb = null_condition_emb


class NativeHandler(NativeDitMixin, NativeVaeMixin):
    """Self-contained counterpart of ``AceStepHandler`` for the denoise + decode path.

    Differences from the reference handler are confined to what is out of scope: conditioning arrives pre-embedded
    (``encoder_hidden_states``, ``context_latents``) instead of captions/lyrics; there is no LM phase.
    """

    sample_rate = 48000  # acestep/handler.py:120

    def __init__(self):
        self.model = None
        self.vae = None
        self.device = "cpu"
        self.dtype = torch.float32
        self.use_lora = False
        self.quantization = None
        self.offload_to_cpu = False
        self.current_offload_cost = 0.0
        self.model_variant = "sft"  # which checkpoint family's generate_audio semantics apply: "base" | "sft" | "turbo"
        # data-parallel requests run in the library's launch-shape-independent mode (`shape_independent` below): a song's bits then do
        # not depend on how many ranks share the request.  False: the default (fastest) launch policy, batch-dependent low bits.
        self.dp_shape_independent = True

    def initialize_service(self, dit_config: DitConfig, decoder_state_dict: Dict[str, torch.Tensor],
                           null_condition_emb: torch.Tensor, vae_config: Optional[VaeConfig] = None,
                           vae_state_dict: Optional[Dict[str, torch.Tensor]] = None, device: str = "auto",
                           use_native: bool = True, model_variant: str = "sft"):
        """Counterpart of ``initialize_service`` (handler/init_service_orchestrator.py:15-110) for pre-loaded weights.
        Re-entry is allowed (:30-31).  Returns ``(status_message, success)`` like the reference.
        ``model_variant`` names the checkpoint family (the reference reads it off the checkpoint directory,
        handler/init_service_catalog.py:69-73): "base" ignores explicit ``timesteps`` like its model, "sft" honours them,
        "turbo" selects the 8-step table sampler without CFG."""
        try:
            if model_variant not in ("base", "sft", "turbo"):
                raise ValueError(f"unknown model_variant '{model_variant}'")
            self.model_variant = model_variant
            if device == "auto":
                device = "cuda" if torch.cuda.is_available() else "cpu"
            if device == "cuda":
                device = f"cuda:{torch.cuda.current_device()}"
            self.device = device
            self.dtype = torch.bfloat16 if str(device).startswith("cuda") else torch.float32  # orchestrator :51
            self.model = _ModelShell(dit_config, decoder_state_dict, null_condition_emb)
            self.vae_config, self.vae_state_dict = vae_config, vae_state_dict
            if not use_native:
                return "native backend disabled", False
            if not str(device).startswith("cuda"):
                raise RuntimeError("the native backend needs a ROCm GPU (no CPU fallback)")
            ok = self._init_native_dit()
            if ok and vae_state_dict is not None:
                ok = self._init_native_vae()
            return ("native backend ready" if ok else "native backend failed to initialise"), bool(ok)
        except Exception as exc:
            logger.exception("initialize_service failed")
            return f"Error: {exc!s}", False

    # ------------------------------------------------------------------ service_generate
    def service_generate(self, encoder_hidden_states: torch.Tensor, context_latents: torch.Tensor,
                         src_latents: Optional[torch.Tensor] = None, seed: Union[int, List[int], None] = None,
                         infer_steps: int = 30, guidance_scale: float = 7.0, audio_cover_strength: float = 1.0,
                         cover_noise_strength: float = 0.0, infer_method: str = "ode", use_adg: bool = False,
                         cfg_interval_start: float = 0.0, cfg_interval_end: float = 1.0, shift: float = 1.0,
                         timesteps: Optional[Sequence[float]] = None, encoder_hidden_states_non_cover=None,
                         context_latents_non_cover=None) -> Dict[str, Any]:
        """The diffusion half of ``service_generate`` (handler/service_generate.py:21-146) from prepared conditions."""
        B = context_latents.shape[0]
        if B > MAX_BATCH_SIZE:  # handler/service_generate_request.py:66-75 clamps; we refuse explicitly
            raise ValueError(f"batch size {B} exceeds the per-call cap of {MAX_BATCH_SIZE}")
        if src_latents is None:
            src_latents = context_latents[..., : context_latents.shape[-1] // 2]
        with torch.inference_mode():
            outputs = self._native_run_diffusion(
                encoder_hidden_states=encoder_hidden_states, encoder_attention_mask=None, context_latents=context_latents,
                src_latents=src_latents, seed=seed, infer_method=infer_method, shift=shift, timesteps=timesteps,
                audio_cover_strength=audio_cover_strength, encoder_hidden_states_non_cover=encoder_hidden_states_non_cover,
                context_latents_non_cover=context_latents_non_cover, infer_steps=infer_steps, guidance_scale=guidance_scale,
                cfg_interval_start=cfg_interval_start, cfg_interval_end=cfg_interval_end, use_adg=use_adg,
                cover_noise_strength=cover_noise_strength)
        outputs.update({"src_latents": src_latents, "encoder_hidden_states": encoder_hidden_states,
                        "context_latents": context_latents, "encoder_attention_mask": None})
        return outputs

    # ------------------------------------------------------------------ generate_music
    def generate_music(self, encoder_hidden_states: Optional[torch.Tensor], context_latents: Optional[torch.Tensor], seed=None,
                       inference_steps: int = 27, guidance_scale: float = 7.0, shift: float = 1.0, infer_method: str = "ode",
                       timesteps=None, use_tiled_decode: bool = True, latent_shift: float = 0.0, latent_rescale: float = 1.0,
                       cfg_interval_start: float = 0.0, cfg_interval_end: float = 1.0, use_adg: bool = False,
                       progress=None, data_parallel: bool = False, **service_kwargs) -> Dict[str, Any]:
        """Counterpart of ``AceStepHandler.generate_music`` (handler/generate_music.py:22-190) from prepared conditions.
        Never raises: every exception becomes the reference's error payload.

        ``data_parallel=True`` inside an initialised ``torch.distributed`` group (one process per GPU, every rank holding an
        initialised handler): rank 0 passes the request of G songs (one seed per song), the other ranks pass ``None`` tensors;
        the request is broadcast once, each rank generates its contiguous slice (``ace355.dist.run_request``: 8/4/2/1 songs
        per rank for a batch of 8 on 1/2/4/8 GPUs) and rank 0 returns the payload with all G songs in order (the other ranks
        return theirs).  Every per-request setting (steps, guidance, shift, CFG interval, ADG, ode / sde, explicit timesteps, tiled
        decode, latent shift / rescale, cover strength / cover noise strength) is rank 0's and travels with the request, and so do a cover
        request's tensors (``src_latents``, ``encoder_hidden_states_non_cover``, ``context_latents_non_cover`` in ``service_kwargs``);
        any other keyword is refused on every rank.  A song's noise, schedule and conditions never depend on the rank or the batch it
        runs in; by default (``self.dp_shape_independent = True``) neither do its bits: the request runs inside ``shape_independent()`` (one K order,
        no split-K / split-KV, one sampler chain), so rank 0's G songs equal a single-process call of the same request under the same mode bit
        for bit (tests/test_dist_gpu.py).  With ``dp_shape_independent = False`` every rank keeps the fastest launch policy for ITS slice and the
        low bits follow the slice size (~3e-3 relative L2, both at the reference's distance); the payload says which:
        ``extra_outputs["data_parallel"]["batch_dependent_bits"]``.  ``extra_outputs`` (pred_latents, src_latents, ...)
        describe the RETURNING rank's slice ``extra_outputs["song_range"]``, ``audios`` on rank 0 all G songs.  The per-call cap of 8 (handler/service_generate_request.py:12) then holds per rank."""
        if data_parallel:
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_world_size() > 1:
                return self._generate_music_data_parallel(
                    encoder_hidden_states, context_latents, seed, progress,
                    dict(inference_steps=inference_steps, guidance_scale=guidance_scale, shift=shift, cfg_interval_start=cfg_interval_start,
                         cfg_interval_end=cfg_interval_end, use_adg=float(bool(use_adg)), infer_method_sde=float(infer_method == "sde")),
                    dict(timesteps=timesteps, use_tiled_decode=use_tiled_decode, latent_shift=latent_shift, latent_rescale=latent_rescale,
                         **service_kwargs))
        try:
            payload, _ = self._generate_music_local(encoder_hidden_states, context_latents, seed, inference_steps, guidance_scale, shift,
                                                    infer_method, timesteps, use_tiled_decode, latent_shift, latent_rescale,
                                                    cfg_interval_start, cfg_interval_end, use_adg, progress, service_kwargs)
            return payload
        except Exception as exc:  # handler/generate_music.py:181-190
            return self._error_payload(exc)

    @staticmethod
    def _error_payload(exc) -> Dict[str, Any]:
        logger.exception("[generate_music] Generation failed")
        return {"audios": [], "status_message": f"Error: {exc!s}\n{traceback.format_exc()}", "extra_outputs": {},
                "success": False, "error": f"{exc!s}"}

    def _generate_music_local(self, encoder_hidden_states, context_latents, seed, inference_steps, guidance_scale, shift, infer_method,
                              timesteps, use_tiled_decode, latent_shift, latent_rescale, cfg_interval_start, cfg_interval_end, use_adg,
                              progress, service_kwargs):
        """The single-GPU body: service_generate -> _prepare_decode_state -> tiled_decode -> peak normalise -> payload.
        Returns (payload, waveforms on the device [B, 2, samples]); raises on failure."""
        if progress:
            progress(0.52, desc="Generating music...")
        outputs = self.service_generate(encoder_hidden_states, context_latents, seed=seed, infer_steps=inference_steps,
                                        guidance_scale=guidance_scale, shift=shift, infer_method=infer_method,
                                        timesteps=timesteps, cfg_interval_start=cfg_interval_start,
                                        cfg_interval_end=cfg_interval_end, use_adg=use_adg, **service_kwargs)
        pred_latents, time_costs = self._prepare_decode_state(outputs, latent_shift, latent_rescale)
        if progress:
            progress(0.8, desc="Decoding audio...")
        t0 = time.time()
        with torch.inference_mode():
            pred_latents_cpu = pred_latents.detach().cpu()
            z = pred_latents.transpose(1, 2).contiguous()  # handler/generate_music_decode.py:123
            pred_wavs = self.tiled_decode(z) if use_tiled_decode else self._native_vae_decode(z)
            if pred_wavs.dtype != torch.float32:
                pred_wavs = pred_wavs.float()
            if pred_wavs.is_cuda:
                from .vae import peak_normalize
                pred_wavs = peak_normalize(pred_wavs.contiguous())
            else:
                peak = pred_wavs.abs().amax(dim=[1, 2], keepdim=True)
                if torch.any(peak > 1.0):
                    pred_wavs = pred_wavs / peak.clamp(min=1.0)
            if pred_wavs.is_cuda:
                torch.cuda.synchronize(pred_wavs.device)
        time_costs["vae_decode_time_cost"] = time.time() - t0
        time_costs["total_time_cost"] = time_costs["total_time_cost"] + time_costs["vae_decode_time_cost"]
        time_costs["offload_time_cost"] = self.current_offload_cost
        B = pred_wavs.shape[0]
        audios = [{"tensor": pred_wavs[i].cpu(), "sample_rate": self.sample_rate} for i in range(B)]
        seed_value = seed[0] if isinstance(seed, (list, tuple)) and seed else seed
        extra = {"pred_latents": pred_latents_cpu, "target_latents": None,
                 "src_latents": outputs["src_latents"].detach().cpu(), "chunk_masks": None, "latent_masks": None, "spans": [],
                 "time_costs": time_costs, "seed_value": seed_value,
                 "encoder_hidden_states": outputs["encoder_hidden_states"].detach().cpu(), "encoder_attention_mask": None,
                 "context_latents": outputs["context_latents"].detach().cpu(), "lyric_token_idss": None}
        return ({"audios": audios, "status_message": "Generation completed successfully!", "extra_outputs": extra,
                 "success": True, "error": None}, pred_wavs)

    def _generate_music_data_parallel(self, encoder_hidden_states, context_latents, seed, progress, knobs, local_kwargs) -> Dict[str, Any]:
        """One request over every rank of the process group (``ace355.dist.run_request``).  A failure on ANY rank becomes the error
        payload on EVERY rank: the ranks agree on success (one MAX all-reduce) before the gather, so nobody waits in a
        collective the failed rank never joins."""
        import torch.distributed as dist
        from . import dist as a_dist
        rank, world = dist.get_rank(), dist.get_world_size()
        err: Optional[BaseException] = None
        request = None
        state: Dict[str, Any] = {}
        try:
            if rank == 0:
                if encoder_hidden_states is None or context_latents is None:
                    raise ValueError("data_parallel: rank 0 must pass the request tensors")
                G = max(encoder_hidden_states.shape[0], context_latents.shape[0])
                if not isinstance(seed, (list, tuple)) or len(seed) != G or any(s is None or int(s) < 0 for s in seed):
                    raise ValueError("data_parallel: one explicit seed per song is required (per-item generators make a song's NOISE "
                                     "independent of the rank and batch it runs in; a scalar seed couples the batch, base.py:1733-1770)")
                if G > MAX_BATCH_SIZE * world:
                    raise ValueError(f"batch size {G} exceeds the per-call cap of {MAX_BATCH_SIZE} on {world} ranks")
                shipped = ("timesteps", "use_tiled_decode", "latent_shift", "latent_rescale", "audio_cover_strength", "cover_noise_strength",
                           "src_latents", "encoder_hidden_states_non_cover", "context_latents_non_cover")
                extra = {kk: v for kk, v in local_kwargs.items() if kk not in shipped and v is not None}
                if extra:
                    # an argument that does not travel would apply on rank 0 only (advisor r3): refused on every rank rather than
                    # silently diverging.  Cover / repaint requests DO travel (round 5, advisor r4): strengths as knobs, tensors as items.
                    raise ValueError(f"data_parallel: these arguments are not part of the broadcast request: {sorted(extra)}")
                for kk in ("audio_cover_strength", "cover_noise_strength"):
                    if local_kwargs.get(kk) is not None:
                        knobs = dict(knobs, **{kk: float(local_kwargs[kk])})
                src_ship = local_kwargs.get("src_latents")
                if src_ship is None and float(local_kwargs.get("cover_noise_strength") or 0.0) > 0.0:
                    # the renoised start t x noise + (1 - t) x src is fp32 host arithmetic on the FULL-precision source half of the context; the
                    # context itself travels in bf16, so the default source is shipped explicitly (advisor r5: same bits as a single-GPU call)
                    src_ship = context_latents[..., : context_latents.shape[-1] // 2].float()
                request = a_dist.pack_request(encoder_hidden_states, context_latents, [int(s) for s in seed], None,
                                              timesteps=local_kwargs.get("timesteps"), use_tiled_decode=float(bool(local_kwargs.get("use_tiled_decode", True))),
                                              latent_shift=float(local_kwargs.get("latent_shift", 0.0)),
                                              latent_rescale=float(local_kwargs.get("latent_rescale", 1.0)),
                                              src_latents=src_ship,
                                              encoder_hidden_states_non_cover=local_kwargs.get("encoder_hidden_states_non_cover"),
                                              context_latents_non_cover=local_kwargs.get("context_latents_non_cover"), **knobs)
        except Exception as exc:
            err = exc
        # (rank 0 could not even build the request: the others learn it from an empty one)
        if rank == 0 and request is None:
            request = {"enc_rows": torch.zeros(0, 1, 1), "enc_index": torch.zeros(0, dtype=torch.int32), "ctx": torch.zeros(1, 1, 2),
                       "seeds": torch.zeros(0, dtype=torch.int64), "knobs": torch.zeros(len(a_dist.KNOBS), dtype=torch.float64),
                       "timesteps": torch.zeros(0)}

        def execute(local):
            k = local["knobs"]
            # every setting comes from the BROADCAST request (rank 0's call), none from this rank's own arguments
            state["shape_independent"] = bool(self.dp_shape_independent)
            with self.shape_independent(self.dp_shape_independent):
                payload, wavs = self._generate_music_local(
                    local["encoder_hidden_states"], local["context_latents"], local["seeds"], int(k["inference_steps"]), k["guidance_scale"],
                    k["shift"], "sde" if k["infer_method_sde"] else "ode", local["timesteps"], bool(k["use_tiled_decode"]),
                    k["latent_shift"], k["latent_rescale"], k["cfg_interval_start"], k["cfg_interval_end"],
                    bool(k["use_adg"]), progress if rank == 0 else None,
                    dict(audio_cover_strength=k["audio_cover_strength"], cover_noise_strength=k["cover_noise_strength"],
                         src_latents=local["src_latents"], encoder_hidden_states_non_cover=local["encoder_hidden_states_non_cover"],
                         context_latents_non_cover=local["context_latents_non_cover"]))
            state["payload"] = payload
            return wavs

        res = None
        try:
            res = a_dist.run_request(request, lambda local: self._dp_guard(execute, local, state), src=0, device=torch.device(self.device), gather=False)
        except Exception as exc:
            err = err or exc
        if err is None and "error" in state:
            err = state["error"]
        cdev = torch.device("cpu") if a_dist.host_staged() else torch.device(self.device)   # (gloo: collectives on host tensors)
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            return self._error_payload(err if err is not None else RuntimeError("generation failed on another rank"))
        wavs = res["local"]
        if wavs is None:   # this rank owns no song
            wavs = torch.empty(0, 2, 0, device=torch.device(self.device))
        # song-major gather to rank 0 (ace355.dist.gather_waveforms: sizes through one all_gather, payloads point to point)
        gathered = a_dist.gather_waveforms(wavs, dst=0)
        payload = state.get("payload") or {"audios": [], "status_message": "Generation completed successfully!", "extra_outputs": {},
                                           "success": True, "error": None}
        payload["extra_outputs"]["song_range"] = res["range"]
        if rank != 0:
            return payload
        audios = list(payload["audios"])
        for r in range(1, world):
            audios += [{"tensor": gathered[r][i].cpu(), "sample_rate": self.sample_rate} for i in range(gathered[r].shape[0])]
        payload["audios"] = audios
        payload["extra_outputs"]["data_parallel"] = {"world": world, "global_batch": res["global_batch"],
                                                     "batch_dependent_bits": not state.get("shape_independent", False)}
        return payload

    @contextlib.contextmanager
    def shape_independent(self, on: bool = True):
        """Run the enclosed native calls with ONE arithmetic per song whatever the launch shape: `ace355_gemm_set_k_rotation(0)` (no K
        rotation, no split-K, no split-KV / key-split attention: include/ace355.h) and one sampler chain (`ace355_dit_set_dual(0)`).  (The folded RMSNorms stay on: their
        row sums are gathered per aligned 128-column group in one fp32 order whatever the GEMM tile)  A song
        generated alone, inside a batch of 8, or on any rank of a data-parallel request then comes out bit for bit the same
        (tests/test_dist_gpu.py, tests/test_metric_shapes_gpu.py).  Costs the small-request optimisations their gain (measured per
        request in DESIGN.md section 14); process-wide while active (the K-rotation mode is a library global)."""
        if not on or self.native_dit is None:
            yield
            return
        from . import native
        prev_k = native.gemm_set_k_rotation(0)
        prev_d = self.native_dit.set_dual(0)
        try:
            yield
        finally:
            native.gemm_set_k_rotation(prev_k)
            self.native_dit.set_dual(prev_d)

    @staticmethod
    def _dp_guard(execute, local, state):
        try:
            return execute(local)
        except Exception as exc:  # reported through the all-reduce of the caller, not raised past the collectives
            state["error"] = exc
            return None

    def _prepare_decode_state(self, outputs: Dict[str, Any], latent_shift: float, latent_rescale: float):
        """handler/generate_music_decode.py:16-96: NaN/Inf and all-zero guards, latent * rescale + shift."""
        pred = outputs["target_latents"]
        time_costs = outputs["time_costs"]
        time_costs["offload_time_cost"] = self.current_offload_cost
        if pred.is_cuda:
            from .vae import latent_check
            bad, zero = latent_check(pred)
        else:
            bad = bool(torch.isnan(pred).any() or torch.isinf(pred).any())
            zero = bool(pred.numel() > 0 and pred.abs().sum() == 0)
        if bad:
            raise RuntimeError("Generation produced NaN or Inf latents. This usually indicates a checkpoint/config mismatch "
                               "or unsupported quantization/backend combination.")
        if zero:
            raise RuntimeError("Generation produced zero latents. This usually indicates a checkpoint/config mismatch or unsupported setup.")
        if latent_shift != 0.0 or latent_rescale != 1.0:
            pred = pred * latent_rescale + latent_shift
        return pred.float(), time_costs
