"""Round 6: where does a song's arithmetic still depend on the batch it runs in, in the launch-shape-independent mode?
One CFG pair alone (N = 2) and inside N = 16 at the metric length, `ace355_gemm_set_k_rotation(0)`: first decoder layer whose residual
stream differs, then one sampler step at B = 1 vs B = 8."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import ace355
    from ace355 import native, weightgen
    from ace355.dit import NativeDit, generate_latents, prepare_noise
    dev = torch.device("cuda:0")
    cfg = ace355.DitConfig()
    dit = NativeDit(cfg, dev)
    for name, shape in cfg.weight_shapes().items():
        wt = weightgen.make_dit_weights({name: shape}, cfg.hidden_size, seed=4, mode="test")[name]
        native.check(dit._lib.ace355_dit_load_tensor(dit._h, name.encode(), native.ptr(wt.contiguous()), 0, wt.numel(), 0), name)
    native.check(dit._lib.ace355_dit_finalize(dit._h), "finalize")
    null = weightgen.make_null_condition_emb(cfg.hidden_size, seed=4)
    g = torch.Generator().manual_seed(3)
    enc = torch.randn(769, cfg.hidden_size, generator=g)
    T, S = 750, 375
    x8 = prepare_noise((8, T, 64), [1000 + i for i in range(8)])
    ctx1 = torch.cat([0.5 * torch.randn(1, T, 64, generator=g), torch.ones(1, T, 64)], -1)
    native.gemm_set_k_rotation(0)
    dit.set_dual(0)
    dit.set_norm_fold(int(os.environ.get("FOLD", "0")))   # FOLD=1: the folded norms stay on (shape-independent since their row sums are per 128-column group)
    dit.set_condition(0, enc)
    dit.set_condition(1, null.reshape(1, -1), L=769)
    t = 0.7
    it = 5

    def fwd(xs, slots, li):
        N = xs.shape[0]
        tap = torch.empty(N * S, cfg.hidden_size, device=dev)
        dit.set_tap(li, tap)
        try:
            v = dit.forward(xs, ctx1.expand(N, -1, -1).contiguous(), [t] * N, [t] * N, slots)
            torch.cuda.synchronize()
        finally:
            dit.set_tap(li, None)
        return v.cpu(), tap.view(N, S, -1).cpu()
    first = None
    for li in (0, 1, 2, 5, 11, 23):
        v2, t2 = fwd(torch.cat([x8[it:it + 1], x8[it:it + 1]]), [0, 1], li)
        v16, t16 = fwd(torch.cat([x8, x8]), [0] * 8 + [1] * 8, li)
        eq_c, eq_u = torch.equal(t2[0], t16[it]), torch.equal(t2[1], t16[8 + it])
        print(f"layer {li:2d} residual stream: cond rows equal {eq_c}, null rows equal {eq_u}" +
              ("" if eq_c and eq_u else f"  (rel {float((t2[0] - t16[it]).norm() / t2[0].norm()):.2e} / {float((t2[1] - t16[8 + it]).norm() / t2[1].norm()):.2e})"), flush=True)
        if first is None and not (eq_c and eq_u):
            first = li
    print("velocity equal:", torch.equal(v2[0], v16[it]), torch.equal(v2[1], v16[8 + it]))
    kw = dict(infer_steps=3, diffusion_guidance_sale=7.0)
    a = generate_latents(dit, null, enc[None].expand(1, -1, -1), ctx1, seed=[1000 + it], **kw)["target_latents"].cpu()
    b = generate_latents(dit, null, enc[None].expand(8, -1, -1), ctx1.expand(8, -1, -1).contiguous(), seed=[1000 + i for i in range(8)], **kw)["target_latents"].cpu()
    print("three sampler steps (fold %s), song alone == in the batch of 8:" % os.environ.get("FOLD", "0"), torch.equal(a[0], b[it]), f"rel {float((a[0] - b[it]).norm() / a[0].norm()):.2e}")


if __name__ == "__main__":
    main()
