"""GPU parity of the audio-token detokenizer through the C ABI, against golden vectors captured
from the imported reference (tests/golden/g8_detokenizer.npz) and against the oracle at the real architecture."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def test_detokenizer_vs_reference_golden(gpu_device, golden_dir):
    import ace355
    from ace355 import weightgen
    from ace355.lmhints import NativeDetokenizer
    G = np.load(f"{golden_dir}/g8_detokenizer.npz")
    cfg = ace355.DetokConfig(hidden_size=256, intermediate_size=768, num_attention_heads=2, num_key_value_heads=1, head_dim=128)
    w = weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=int(G["seed"]), mode="test")
    assert abs(weightgen.checksum(w) - float(G["wsum"])) < 1e-6 * float(G["wsum"])
    det = NativeDetokenizer(cfg, gpu_device)
    det.load_state_dict(w)
    y = det(torch.from_numpy(G["x"]))
    ref = torch.from_numpy(G["y"])
    assert y.shape == ref.shape
    r = _rel(y.cpu(), ref)
    print(f"detokenizer (tiny) vs reference fp32: rel L2 {r:.3e}")
    assert r < 1.5e-2, r   # reference fp32 CPU vs bf16 kernels, 2 layers


def test_detokenizer_full_size_vs_oracle_and_code_path(gpu_device):
    """Real architecture (2 layers, 2048 wide) on a 10 s clip's worth of codes (50 tokens -> 250 frames), fed through the
    whole N2 chain: code string -> indices -> FSQ output -> detokenizer."""
    import ace355
    from ace355 import weightgen
    from ace355.lmhints import NativeDetokenizer, decode_audio_codes_to_latents, parse_audio_code_string
    from oracle import detok as o_detok
    cfg = ace355.DetokConfig()
    w = weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=12, mode="test")
    det = NativeDetokenizer(cfg, gpu_device)
    det.load_state_dict(w)
    g = torch.Generator().manual_seed(6)
    codes = torch.randint(0, 64000, (50,), generator=g).tolist()
    s = "".join(f"<|audio_code_{c}|>" for c in codes)
    assert parse_audio_code_string(s) == codes == o_detok.parse_audio_code_string(s)
    pw = 0.5 * torch.randn(cfg.hidden_size, 6, generator=g)
    pb = 0.1 * torch.randn(cfg.hidden_size, generator=g)
    y = decode_audio_codes_to_latents(s, det, pw.to(gpu_device), pb.to(gpu_device))
    q = o_detok.fsq_output_from_indices(torch.tensor(codes).view(1, -1, 1), o_detok.FSQ_LEVELS if hasattr(o_detok, "FSQ_LEVELS") else (8, 8, 8, 5, 5, 5), pw, pb)
    ref = o_detok.detokenizer(o_detok.DetokConfig(), w, q)
    assert y.shape == (1, 250, 64) == tuple(ref.shape)
    r = _rel(y.cpu(), ref)
    print(f"detokenizer (full size) vs fp32 oracle: rel L2 {r:.3e}")
    assert r < 1.5e-2, r   # measured 5.6e-3
    assert decode_audio_codes_to_latents("no codes here", det, pw, pb) is None
    with pytest.raises(ValueError):
        det(torch.zeros(1, 4, cfg.hidden_size), attention_mask=torch.ones(1, 4))
