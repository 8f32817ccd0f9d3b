"""GPU parity of the individual HIP kernels (through the C ABI test hooks) against the oracle / plain fp32 torch.

Inputs are rounded to bf16 first, so the comparison isolates the kernel's arithmetic (fp32 accumulate,
bf16 output rounding) from input quantisation.  Tolerances are written next to each check.
"""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16)


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


@pytest.fixture(scope="module")
def lib(gpu_device):
    from ace355 import native
    return native.lib()


def _p(t):
    from ace355 import native
    return native.ptr(t)


def _chk(rc):
    from ace355 import native
    native.check(rc, "test")
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (6000, 2048, 2048), (130, 512, 384), (1, 256, 256), (257, 128, 6144), (750, 12288, 2048)])
def test_gemm_store(lib, gpu_device, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    # asymmetric, non-symmetric operands (transpose-detecting, cdna guide 5.4 rule 16)
    A = _bf(torch.randn(M, K, generator=g)).to(gpu_device)
    W = _bf(torch.randn(N, K, generator=g) * 0.05 + torch.arange(N)[:, None] * 1e-3).to(gpu_device)
    bias = torch.randn(N, generator=g).to(gpu_device)
    ref = A.float() @ W.float().t() + bias
    out = torch.empty(M, N, device=gpu_device, dtype=torch.float32)
    _chk(lib.ace355_gemm_bf16(_p(A), _p(W), _p(out), M, N, K, 0, _p(bias), None))
    assert _rel(out, ref) < 2e-5, _rel(out, ref)  # fp32 accumulate of exact bf16 products, f32 store
    outb = torch.empty(M, N, device=gpu_device, dtype=torch.bfloat16)
    _chk(lib.ace355_gemm_bf16(_p(A), _p(W), _p(outb), M, N, K, 1, None, None))
    ref2 = (A.float() @ W.float().t())
    assert _rel(outb, ref2) < 4e-3  # bf16 output rounding (2^-9 relative)
    assert torch.equal(outb, ref2.to(torch.bfloat16)) or float((outb.float() - ref2).abs().max() / ref2.abs().max()) < 8e-3


def test_gemm_k_rotation_modes(lib, gpu_device):
    """include/ace355.h ace355_gemm_set_k_rotation: the one-round launches walk K rotated by (XCD of the tile) * nk / 8.  Every mode must
    give the same product up to the fp32 summation order (every K slice visited exactly once from every starting point), the rotated
    order must really differ from the plain one on a chip-filling launch, and each mode is bit-reproducible."""
    from ace355 import native
    M, N, K = 6000, 2048, 2048   # 256 tiles of 192x256: one round, every XCD its own rotation
    g = torch.Generator().manual_seed(77)
    A = _bf(torch.randn(M, K, generator=g)).to(gpu_device)
    W = _bf(torch.randn(N, K, generator=g) * 0.05 + torch.arange(N)[:, None] * 1e-3).to(gpu_device)
    ref = A.float() @ W.float().t()
    outs = {}
    prev = native.gemm_set_k_rotation(0)
    try:
        for mode in (0, 1, 2):
            native.gemm_set_k_rotation(mode)
            o1 = torch.empty(M, N, device=gpu_device, dtype=torch.float32)
            o2 = torch.empty_like(o1)
            _chk(lib.ace355_gemm_bf16(_p(A), _p(W), _p(o1), M, N, K, 0, None, None))
            _chk(lib.ace355_gemm_bf16(_p(A), _p(W), _p(o2), M, N, K, 0, None, None))
            assert torch.equal(o1, o2), f"mode {mode} is not reproducible"
            assert _rel(o1, ref) < 2e-5, (mode, _rel(o1, ref))
            outs[mode] = o1
        assert native.gemm_set_k_rotation(prev) == 2
    finally:
        native.gemm_set_k_rotation(prev)
    d01 = _rel(outs[1], outs[0])
    print(f"K rotation: mode 1 vs mode 0 rel L2 {d01:.2e} (summation order only), mode 2 vs mode 1 {_rel(outs[2], outs[1]):.2e}")
    assert 0.0 < d01 < 1e-6, d01          # a different fp32 order (not bit-identical), the same sum


@pytest.mark.parametrize("M,N,K,rows", [(375 * 4, 2048, 2048, 375), (60, 256, 256, 20), (130, 256, 768, 65)])
def test_gemm_residual_gate(lib, gpu_device, M, N, K, rows):
    g = torch.Generator().manual_seed(7)
    A = _bf(torch.randn(M, K, generator=g)).to(gpu_device)
    W = _bf(torch.randn(N, K, generator=g) * 0.05).to(gpu_device)
    H = torch.randn(M, N, generator=g).to(gpu_device)
    nseq = M // rows + (1 if M % rows else 0)
    g1 = torch.randn(N, generator=g).to(gpu_device)
    g2 = torch.randn(nseq, 6, N, generator=g).to(gpu_device)  # stride 6*N like timestep_proj
    seq = torch.arange(M, device=gpu_device) // rows
    gate = g1[None, :] + g2[seq, 2, :]
    ref = H + gate * (A.float() @ W.float().t())
    out = H.clone()
    _chk(lib.ace355_gemm_bf16_fused(_p(A), _p(W), _p(out), M, N, K, 0, _p(g1), g2[:, 2].data_ptr(), 6 * N, rows, None))
    assert _rel(out, ref) < 2e-5, _rel(out, ref)
    out2 = H.clone()  # plain residual (cross-attention, base.py:526)
    _chk(lib.ace355_gemm_bf16_fused(_p(A), _p(W), _p(out2), M, N, K, 0, None, None, 0, rows, None))
    assert _rel(out2, H + A.float() @ W.float().t()) < 2e-5


# ---- the instantiations the headline runs (profiles/*_bench_kernel_stats.csv): M = 6000 = 16 x 375 token rows (192x256 persistent
# tiles), Mc = 3000 cross-attention rows (192x128 mid tile), and the batch-1 regime M = 750 / 375 (4-wave deep-pipeline tiles,
# split-K residuals).  Reference = plain fp32 matmul of the same bf16 operands on the GPU.
@pytest.mark.parametrize("M,N,K,rows,cvec_row0", [
    (6000, 2048, 2048, 375, 3000),   # o_proj: gemm_sp_kernel<2,3,4,2,0,2>-class launch (one round of 256 tiles) + constant null term
    (6000, 2048, 6144, 375, -1),     # down_proj (K = 6144)
    (3000, 2048, 2048, 375, -1),     # cross-attention o_proj: 192x128 mid tile, plain residual
    (750, 2048, 2048, 375, 375),     # batch 1: split-K over blockIdx.y in part order
    (750, 2048, 6144, 375, -1),
    (375, 2048, 2048, 375, -1)])
def test_gemm_residual_metric_shapes(lib, gpu_device, M, N, K, rows, cvec_row0):
    g = torch.Generator().manual_seed(M + K)
    A = _bf(torch.randn(M, K, generator=g)).to(gpu_device)
    W = _bf(torch.randn(N, K, generator=g) * 0.02).to(gpu_device)
    H = torch.randn(M, N, generator=g).to(gpu_device)
    nseq = (M + rows - 1) // rows
    gated = cvec_row0 != -1 or K == 6144
    g1 = torch.randn(N, generator=g).to(gpu_device)
    g2 = torch.randn(nseq, 6, N, generator=g).to(gpu_device)
    cvec = torch.randn(N, generator=g).to(gpu_device) if cvec_row0 >= 0 else None
    seq = torch.arange(M, device=gpu_device) // rows
    prod = A.float() @ W.float().t()
    ref = H + ((g1[None] + g2[seq, 5]) * prod if gated else prod)
    if cvec is not None:
        ref[cvec_row0:] += cvec
    out = H.clone()
    _chk(lib.ace355_gemm_bf16_residual(_p(A), _p(W), _p(out), M, N, K, _p(g1) if gated else None,
                                       g2[:, 5].data_ptr() if gated else None, 6 * N, rows, _p(cvec), max(cvec_row0, 0), None))
    r = _rel(out - H, ref - H)  # on the UPDATE, so that H does not mask an error in it
    print(f"residual GEMM M={M} N={N} K={K}: rel L2 of the update {r:.2e}")
    assert r < 2.5e-6, r  # measured 3.8e-7 - 8.6e-7 (fp32 accumulation order only; the split-K parts included)
    # the small-M launches split K over two workgroups per tile: the parts add in part order (turn counters), so a repeat is bit-identical
    for _ in range(3):
        again = H.clone()
        _chk(lib.ace355_gemm_bf16_residual(_p(A), _p(W), _p(again), M, N, K, _p(g1) if gated else None,
                                           g2[:, 5].data_ptr() if gated else None, 6 * N, rows, _p(cvec), max(cvec_row0, 0), None))
        assert torch.equal(again, out), "residual GEMM is not bit-reproducible"


@pytest.mark.parametrize("M,N,K,q_cols,qk_cols,rope,rows", [
    (6000, 4096, 2048, 2048, 3072, 1, 375),   # self-attention QKV: persistent 192x256, head-norm + RoPE on q / k, v passes through
    (3000, 2048, 2048, 2048, 2048, 0, 375),   # cross-attention q: 192x128 mid tile, head-norm only
    (750, 4096, 2048, 2048, 3072, 1, 375),    # batch 1: 4-wave deep-pipeline tile with the same epilogue
    (375, 2048, 2048, 2048, 2048, 0, 375)])
def test_gemm_headnorm_rope_metric_shapes(lib, gpu_device, M, N, K, q_cols, qk_cols, rope, rows):
    from oracle import dit as o_dit
    g = torch.Generator().manual_seed(M + N)
    A = _bf(torch.randn(M, K, generator=g)).to(gpu_device)
    W = _bf(torch.randn(N, K, generator=g) * 0.02).to(gpu_device)
    wq = (1 + 0.1 * torch.randn(128, generator=g)).to(gpu_device)
    wk = (1 + 0.1 * torch.randn(128, generator=g)).to(gpu_device)
    y = (A.float() @ W.float().t())
    ref = y.clone()
    cos, sin = o_dit.rope_cos_sin(rows, 128, 1e6)
    pos = torch.arange(M, device=gpu_device) % rows
    cos, sin = cos.to(gpu_device)[pos][:, None], sin.to(gpu_device)[pos][:, None]  # [M,1,128]
    for c0, c1, w in ((0, q_cols, wq), (q_cols, qk_cols, wk)):
        if c1 == c0:
            continue
        h = o_dit.rms_norm(y[:, c0:c1].reshape(M, -1, 128), w, 1e-6)
        if rope:
            h = h * cos + o_dit.rotate_half(h) * sin
            h = torch.stack([h[..., :64], h[..., 64:]], -1).reshape(M, -1, 128)  # library order: dims (d, d+64) adjacent
        ref[:, c0:c1] = h.reshape(M, -1)
    out = torch.empty(M, N, device=gpu_device, dtype=torch.bfloat16)
    _chk(lib.ace355_gemm_bf16_headnorm(_p(A), _p(W), _p(out), M, N, K, q_cols, qk_cols, _p(wq), _p(wk), 1e-6, rope, rows, 1e6, None))
    r_qk, r_v = _rel(out[:, :qk_cols], ref[:, :qk_cols]), (_rel(out[:, qk_cols:], ref[:, qk_cols:]) if N > qk_cols else 0.0)
    print(f"head-norm GEMM M={M} N={N} rope={rope}: rel L2 q/k {r_qk:.2e}, v {r_v:.2e}")
    assert r_qk < 5e-3 and r_v < 4e-3, (r_qk, r_v)  # one or two bf16 roundings (2^-9 each)


@pytest.mark.parametrize("M,Fh,K", [(300, 768, 256), (1000, 6144, 2048), (6000, 6144, 2048), (750, 6144, 2048)])
def test_gemm_swiglu(lib, gpu_device, M, Fh, K):
    g = torch.Generator().manual_seed(9)
    A = _bf(torch.randn(M, K, generator=g)).to(gpu_device)
    Wg = _bf(torch.randn(Fh, K, generator=g) * 0.05).to(gpu_device)
    Wu = _bf(torch.randn(Fh, K, generator=g) * 0.05).to(gpu_device)
    # library layout: rows interleaved [32 gate | 32 up]
    Wp = torch.stack([Wg.view(Fh // 32, 32, K), Wu.view(Fh // 32, 32, K)], dim=1).reshape(2 * Fh, K).contiguous()
    ref = F.silu(A.float() @ Wg.float().t()) * (A.float() @ Wu.float().t())
    out = torch.empty(M, Fh, device=gpu_device, dtype=torch.bfloat16)
    _chk(lib.ace355_gemm_bf16_fused(_p(A), _p(Wp), _p(out), M, 2 * Fh, K, 1, None, None, 0, 0, None))
    assert _rel(out, ref) < 4e-3, _rel(out, ref)


@pytest.mark.parametrize("M,D,rows,mod", [(750, 2048, 375, True), (41, 256, 21, True), (64, 2048, 64, False)])
def test_rmsnorm_mod(lib, gpu_device, M, D, rows, mod):
    from oracle import dit as o_dit
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(M, D, generator=g) * 3).to(gpu_device)
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(gpu_device)
    nseq = (M + rows - 1) // rows
    t1 = torch.randn(6, D, generator=g).to(gpu_device) / D ** 0.5
    t2 = torch.randn(nseq, 6, D, generator=g).to(gpu_device)
    ref = o_dit.rms_norm(x.cpu(), w.cpu(), 1e-6)
    if mod:
        seq = torch.arange(M) // rows
        ref = ref * (1 + (t1[1].cpu() + t2[seq, 1].cpu())) + (t1[0].cpu() + t2[seq, 0].cpu())
    y = torch.empty(M, D, device=gpu_device, dtype=torch.bfloat16)
    if mod:
        _chk(lib.ace355_rmsnorm_mod(_p(x), _p(w), _p(y), M, D, 1e-6, t1[1].data_ptr(), t2[:, 1].data_ptr(), t1[0].data_ptr(),
                                    t2[:, 0].data_ptr(), 6 * D, rows, None))
    else:
        _chk(lib.ace355_rmsnorm_mod(_p(x), _p(w), _p(y), M, D, 1e-6, None, None, None, None, 0, rows, None))
    assert _rel(y.cpu(), ref) < 3e-3  # bf16 output rounding


@pytest.mark.parametrize("N,S,heads,rope", [(2, 375, 16, True), (1, 20, 3, True), (2, 77, 8, False)])
def test_headnorm_rope(lib, gpu_device, N, S, heads, rope):
    from oracle import dit as o_dit
    g = torch.Generator().manual_seed(5)
    M = N * S
    ld = heads * 128 + 256
    x = _bf(torch.randn(M, ld, generator=g) * 2)
    w = 1 + 0.1 * torch.randn(128, generator=g)
    xh = x[:, 128:128 + heads * 128].float().view(N, S, heads, 128)
    ref = o_dit.rms_norm(xh, w, 1e-6).transpose(1, 2)  # [N,H,S,128]
    if rope:
        cos, sin = o_dit.rope_cos_sin(S, 128, 1e6)
        ref, _ = o_dit.apply_rope(ref, ref, cos, sin)
    xd = x.clone().to(gpu_device)
    wd = w.to(gpu_device)
    _chk(lib.ace355_headnorm_rope(_p(xd), M, ld, 128, heads, _p(wd), 1e-6, 1 if rope else 0, S, 1e6, None))
    got = xd.cpu()[:, 128:128 + heads * 128].float().view(N, S, heads, 128).transpose(1, 2)
    assert _rel(got, ref) < 3e-3
    # columns outside the head range are untouched
    assert torch.equal(xd.cpu()[:, :128], x[:, :128]) and torch.equal(xd.cpu()[:, 128 + heads * 128:], x[:, 128 + heads * 128:])


@pytest.mark.parametrize("N,Sq,Skv,Hq,Hkv,window", [
    (2, 375, 375, 16, 8, -1), (2, 375, 375, 16, 8, 128), (1, 300, 300, 2, 1, 16), (2, 375, 769, 16, 8, -1),
    (1, 20, 33, 2, 1, -1), (1, 1500, 1500, 4, 2, 128), (1, 129, 129, 2, 2, 128), (1, 64, 1, 2, 1, -1),
    # Hq = 2 Hkv takes attn_gqa_kernel<NWH> once its grid fills the chip (launch_attention): NWH = 6 (12 waves, 192-row blocks:
    # configs[2]'s self-attention, full and banded, with a ragged last block 1500 = 7 x 192 + 156, and its cross-attention
    # shape), the metric's launches at N = 16 (NWH = 6: 375 = 192 + 183) and the metric's cross-attention at Nc = 8 (NWH = 3)
    (6, 1500, 1500, 16, 8, -1), (6, 1500, 1500, 16, 8, 128), (6, 1500, 769, 16, 8, -1),
    (16, 375, 375, 16, 8, -1), (16, 375, 375, 16, 8, 128), (8, 375, 769, 16, 8, -1), (16, 375, 769, 16, 8, -1),
    (12, 375, 375, 16, 8, 128),   # NWH = 4 (8 waves, 128-row blocks)
    # other group sizes stay on attn3_kernel: <8> (256-query blocks; Sq >= 1024 and heads x ceil(Sq / 256) >= 512) and <4>
    (6, 1500, 1500, 16, 16, -1), (6, 1500, 1500, 16, 16, 128), (4, 375, 375, 8, 8, 128),
    # batch-1 requests: few workgroups with long key walks (>= 8 tiles) take the split-KV path of attn3_kernel<4> + attn_merge_kernel
    # ((2, 375, 769) above: 2 parts): the cross-attention of one conditional sequence (48 workgroups x 13 tiles -> 4 parts),
    # configs[0]'s shapes (Sq = 125: 16 workgroups -> 4 parts; banded self-attention, not split), a ragged split (9 tiles in 4 parts)
    (1, 375, 769, 16, 8, -1), (1, 125, 769, 16, 8, -1), (2, 125, 125, 16, 8, 16), (1, 200, 550, 4, 2, -1)])
def test_attention(lib, gpu_device, N, Sq, Skv, Hq, Hkv, window):
    from oracle import dit as o_dit
    g = torch.Generator().manual_seed(Sq + Skv + Hq)
    q = _bf(torch.randn(N, Sq, Hq * 128, generator=g))
    k = _bf(torch.randn(N, Skv, Hkv * 128, generator=g))
    v = _bf(torch.randn(N, Skv, Hkv * 128, generator=g) + torch.arange(Skv)[None, :, None] * 0.01)
    scale = 128 ** -0.5
    mask = None
    if window >= 0:
        mask = o_dit.additive_mask(o_dit.band_valid(Sq, window))[None, None]
    ref = o_dit.attention(q.float().view(N, Sq, Hq, 128).transpose(1, 2), k.float().view(N, Skv, Hkv, 128).transpose(1, 2),
                          v.float().view(N, Skv, Hkv, 128).transpose(1, 2), mask, scale)
    out = torch.empty(N, Sq, Hq * 128, device=gpu_device, dtype=torch.bfloat16)
    qd, kd, vd = q.to(gpu_device), k.to(gpu_device), v.to(gpu_device)
    _chk(lib.ace355_attention(_p(qd), _p(kd), _p(vd), _p(out), N, Sq, Skv, Hq, Hkv, window, scale, None))
    # P is rounded to bf16 before PV (like every flash kernel) and O to bf16: 1e-2 relative L2
    assert _rel(out.cpu(), ref) < 1e-2, _rel(out.cpu(), ref)


def test_attention_rescale_branch(lib, gpu_device):
    """A key spike late in the sequence forces the online-softmax rescale (cdna guide 5.4 rule 26)."""
    from oracle import dit as o_dit
    g = torch.Generator().manual_seed(1)
    N, S, H = 1, 320, 2
    q = torch.randn(N, S, H * 128, generator=g)
    k = torch.randn(N, S, H * 128, generator=g)
    v = torch.randn(N, S, H * 128, generator=g)
    k[0, 250] = q[0, 7] * 3.0  # huge score for query 7 at key 250 (4th tile)
    q, k, v = _bf(q), _bf(k), _bf(v)
    ref = o_dit.attention(q.float().view(N, S, H, 128).transpose(1, 2), k.float().view(N, S, H, 128).transpose(1, 2),
                          v.float().view(N, S, H, 128).transpose(1, 2), None, 128 ** -0.5)
    out = torch.empty(N, S, H * 128, device=gpu_device, dtype=torch.bfloat16)
    qd, kd, vd = q.to(gpu_device), k.to(gpu_device), v.to(gpu_device)
    _chk(lib.ace355_attention(_p(qd), _p(kd), _p(vd), _p(out), N, S, S, H, H, -1, 128 ** -0.5, None))
    assert float((out.cpu().float() - ref).abs().max()) < 5e-2
    assert _rel(out.cpu(), ref) < 1e-2


@pytest.mark.parametrize("B,T", [(2, 50), (8, 750), (1, 7)])
def test_apg_euler(lib, gpu_device, B, T):
    from oracle import apg as o_apg
    g = torch.Generator().manual_seed(B * T)
    xt = torch.randn(B, T, 64, generator=g)
    x_ref = xt.clone()
    mb = o_apg.MomentumBuffer()
    xd = xt.to(gpu_device)
    avg = torch.zeros(B, T, 64, device=gpu_device)
    for i in range(3):
        v = torch.randn(2 * B, T, 64, generator=g) * (1 + 2 * i)  # growing norms exercise the 2.5 clip
        dt = 0.05 * (i + 1)
        vv = o_apg.apg_forward(v[:B], v[B:], 7.0, mb, dims=[1])
        x_ref = x_ref - vv * dt
        vd = v.to(gpu_device)
        _chk(lib.ace355_apg_euler_step(_p(vd), _p(avg), _p(xd), B, T, 7.0, dt, 1, 1 if i == 0 else 0, None))
        assert float((xd.cpu() - x_ref).abs().max()) < 2e-5 * (1 + float(x_ref.abs().max())), i
    # outside the cfg interval: v = cond, momentum untouched (base.py:1965-1966)
    v = torch.randn(2 * B, T, 64, generator=g)
    avg_before = avg.clone()
    x_ref = x_ref - v[:B] * 0.1
    vd = v.to(gpu_device)
    _chk(lib.ace355_apg_euler_step(_p(vd), _p(avg), _p(xd), B, T, 7.0, 0.1, 0, 0, None))
    assert float((xd.cpu() - x_ref).abs().max()) < 2e-5 * (1 + float(x_ref.abs().max()))
    assert torch.equal(avg, avg_before)


@pytest.mark.parametrize("B,L,Cin,Cout,taps,dil,snake,res", [
    (2, 300, 128, 128, 7, 1, True, False), (1, 517, 128, 128, 7, 9, True, False), (2, 200, 256, 256, 7, 3, True, False),
    (2, 300, 128, 128, 1, 1, True, True), (1, 40, 64, 2048, 7, 1, False, False), (1, 1000, 128, 2, 7, 1, True, False),
    # >= 4096 workgroups at Cin >= 256: the launcher picks the 8-wave 256-row tile (ragged last tile: 66000 % 256 = 208)
    (4, 66000, 256, 256, 7, 3, True, True)])
def test_conv1d_nlc(lib, gpu_device, B, L, Cin, Cout, taps, dil, snake, res):
    from oracle import oobleck as o_vae
    g = torch.Generator().manual_seed(L + Cin + taps + dil)
    x = _bf(torch.randn(B, L, Cin, generator=g))
    w = _bf(torch.randn(Cout, Cin, taps, generator=g) / (Cin * taps) ** 0.5)
    bias = torch.randn(Cout, generator=g) * 0.1
    alpha = torch.randn(Cin, generator=g) * 0.3
    beta = torch.randn(Cin, generator=g) * 0.3
    r = _bf(torch.randn(B, L, Cout, generator=g))
    xin = x.float().transpose(1, 2)
    if snake:
        # the kernel rounds snake(x) to bf16 before the MFMA, so does the reference here
        xin = _bf(o_vae.snake(xin, alpha.view(1, -1, 1), beta.view(1, -1, 1))).float()
    ref = F.conv1d(xin, w.float(), bias, dilation=dil, padding=(taps // 2) * dil).transpose(1, 2)
    if res:
        ref = ref + r.float()
    y = torch.empty(B, L, Cout, device=gpu_device, dtype=torch.bfloat16)
    wp = w.permute(0, 2, 1).contiguous().to(gpu_device)  # [Cout][taps][Cin]
    xd, bd, ad, btd, rd = x.to(gpu_device), bias.to(gpu_device), alpha.to(gpu_device), beta.to(gpu_device), r.to(gpu_device)
    _chk(lib.ace355_conv1d_nlc(_p(xd), _p(wp), _p(bd), _p(ad) if snake else None, _p(btd) if snake else None,
                               _p(rd) if res else None, _p(y), B, L, Cin, Cout, taps, dil, None))
    # fast sin + bf16 rounding of snake(x) may flip a bf16 ulp on a few inputs: 6e-3 relative L2
    assert _rel(y.cpu(), ref) < 6e-3, _rel(y.cpu(), ref)


def test_gemm_tile_forms_give_the_same_bits(gpu_device, tmp_path):
    """the order of the MFMAs inside a K step is a matter of energy, not of arithmetic - every accumulator still takes K half P, then
    K half Q, step after step.  So the three K loops of gemm.hip must agree BIT FOR BIT on one problem once the launch-shape-dependent K rotation is off:
    the 4-wave tiles (half-by-half loop: ACE355_GEMM_BIG=0), the 8-wave 192x256 tile (kstep_pair: =2) and the 8-wave 192x128 tile (kstep_pair1: =3).
    Fresh processes: the switches are read once."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "import ace355\n"
        "from ace355 import native\n"
        "lib = native.lib(); P = native.ptr; dev = torch.device('cuda:0')\n"
        "native.gemm_set_k_rotation(0)\n"
        "g = torch.Generator().manual_seed(11)\n"
        "outs = []\n"
        "for M, N, K in ((6000, 2048, 2048), (1536, 2048, 6144)):\n"
        "    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev); W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)\n"
        "    o = torch.empty(M, N, device=dev)\n"
        "    native.check(lib.ace355_gemm_bf16(P(A), P(W), P(o), M, N, K, 0, None, None)); torch.cuda.synchronize()\n"
        "    ref = A.float() @ W.float().t()\n"
        "    assert float((o - ref).norm() / ref.norm()) < 2e-5\n"
        "    outs.append(o.cpu())\n"
        "torch.save(outs, sys.argv[1])\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    res = {}
    for flag in ("0", "2", "3"):
        path = str(tmp_path / f"big{flag}.pt")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, ACE355_GEMM_BIG=flag, ACE355_GEMM_CLK="0"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[flag] = torch.load(path)
    for flag in ("2", "3"):
        for a, b in zip(res["0"], res[flag]):
            assert torch.equal(a, b), f"ACE355_GEMM_BIG={flag} differs from the 4-wave tiles: max abs {float((a - b).abs().max()):.3e}"


@pytest.mark.parametrize("kind,M,N,K,reps", [
    # the 4-wave two-stage kernels with two workgroups per CU (more tiles than CUs, too few for the 8-wave forms): the regime whose launches were not
    # reproducible until round 6 (MFMA destination overlapping srcB; consecutive MFMAs sharing srcA beside a co-resident bf16 epilogue)
    ("swiglu", 400, 12288, 2048, 16), ("swiglu", 288, 12288, 2048, 16), ("store_bf16", 520, 8192, 2048, 16), ("headnorm", 520, 8192, 2048, 16),
    ("store_f32", 520, 8192, 2048, 8), ("residual", 520, 8192, 2048, 8),
    # deep-pipeline 4-wave kernels (one workgroup per CU), 64-row tiles, ordered split-K
    ("swiglu", 48, 12288, 2048, 8), ("headnorm", 400, 4096, 2048, 8), ("residual", 400, 2048, 6144, 8), ("residual", 375, 2048, 2048, 8),
    # the 8-wave pair loops at the metric batch (persistent 192x256, one-round 192x256, 192x128)
    ("swiglu", 6000, 12288, 2048, 4), ("headnorm", 6000, 4096, 2048, 4), ("residual", 6000, 2048, 6144, 4), ("store_bf16", 400, 12288, 2048, 8),
])
def test_gemm_launches_are_bit_reproducible(lib, gpu_device, kind, M, N, K, reps):
    """Every epilogue mode in every tile regime of launch_gemm, `reps` launches on the same operands: ONE result (the
    condition encoder's gate|up launch returned different bits on every call and nothing in the suite launched that instantiation with a long K loop),
    and the result is the right one (fp32 torch)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = _bf(torch.randn(M, K, generator=g)).to(gpu_device)
    W = _bf(torch.randn(N, K, generator=g) * 0.05).to(gpu_device)
    wq = torch.ones(128, device=gpu_device)
    outs = []
    for _ in range(reps):
        if kind == "store_f32":
            out = torch.empty(M, N, device=gpu_device)
            _chk(lib.ace355_gemm_bf16(_p(A), _p(W), _p(out), M, N, K, 0, None, None))
        elif kind == "store_bf16":
            out = torch.empty(M, N, device=gpu_device, dtype=torch.bfloat16)
            _chk(lib.ace355_gemm_bf16(_p(A), _p(W), _p(out), M, N, K, 1, None, None))
        elif kind == "swiglu":
            out = torch.empty(M, N // 2, device=gpu_device, dtype=torch.bfloat16)
            _chk(lib.ace355_gemm_bf16_fused(_p(A), _p(W), _p(out), M, N, K, 1, None, None, 0, 0, None))
        elif kind == "residual":
            out = torch.ones(M, N, device=gpu_device)
            _chk(lib.ace355_gemm_bf16_residual(_p(A), _p(W), _p(out), M, N, K, None, None, 0, M, None, 0, None))
        else:
            out = torch.empty(M, N, device=gpu_device, dtype=torch.bfloat16)
            _chk(lib.ace355_gemm_bf16_headnorm(_p(A), _p(W), _p(out), M, N, K, N // 2, N // 4 * 3, _p(wq), _p(wq), 1e-6, 1, M, 1e6, None))
        outs.append(out)
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), f"{kind} M={M} N={N} K={K}: launches differ by {_rel(o, outs[0]):.3e}"
    if kind in ("store_f32", "store_bf16"):
        assert _rel(outs[0], A.float() @ W.float().t()) < (2e-5 if kind == "store_f32" else 4e-3)
    elif kind == "residual":
        assert _rel(outs[0] - 1.0, A.float() @ W.float().t()) < 2e-5
    elif kind == "swiglu":
        Wv = W.view(N // 64, 2, 32, K)
        ref = F.silu(A.float() @ Wv[:, 0].reshape(N // 2, K).float().t()) * (A.float() @ Wv[:, 1].reshape(N // 2, K).float().t())
        assert _rel(outs[0], ref) < 4e-3
