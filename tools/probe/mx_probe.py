#!/usr/bin/env python
"""Find the operand / scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 by experiment (no ISA document on the box)."""
import ctypes as C, itertools, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libmx_probe.so"))
lib.mx_probe.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M = N = 32; K = 64
A = (torch.randn(M, K, generator=g)).to(torch.float8_e4m3fn)        # [i][k]
B = (torch.randn(K, N, generator=g)).to(torch.float8_e4m3fn)        # [k][j]
sA = torch.randint(124, 131, (M, 2), generator=g)                    # E8M0 exponent per (row, 32-block)
sB = torch.randint(124, 131, (N, 2), generator=g)
Af, Bf = A.float(), B.float()
ref = torch.zeros(M, N)
for kb in range(2):
    ref += (Af[:, 32 * kb:32 * kb + 32] * (2.0 ** (sA[:, kb].float() - 127))[:, None]) @ (Bf[32 * kb:32 * kb + 32] * (2.0 ** (sB[:, kb].float() - 127))[None, :])
ref_noscale = Af @ Bf
Ab = A.view(torch.uint8); Bb = B.view(torch.uint8)

def run(a_img, b_img, sa_img, sb_img, opa=0, opb=0):
    a = a_img.contiguous().view(torch.int32).to(dev); b = b_img.contiguous().view(torch.int32).to(dev)
    sa = sa_img.to(torch.int32).to(dev); sb = sb_img.to(torch.int32).to(dev)
    c = torch.zeros(64, 16, device=dev)
    rc = lib.mx_probe(a.data_ptr(), b.data_ptr(), sa.data_ptr(), sb.data_ptr(), c.data_ptr(), opa, opb)
    assert rc == 0, rc
    c = c.cpu()
    D = torch.zeros(32, 32)  # standard 32x32 C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    for l in range(64):
        for r in range(16):
            D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = c[l, r]
    return D

lanes = torch.arange(64)
def img_contig(Mb, row_major_k):     # lane l: row l&31, k = 32*(l>>5) + 0..31
    out = torch.zeros(64, 32, dtype=torch.uint8)
    for l in range(64):
        out[l] = row_major_k[l & 31, 32 * (l >> 5):32 * (l >> 5) + 32]
    return out
def img_inter16(Mb, rk):             # lane l: k = 16*(l>>5) + 0..15 and 32 + 16*(l>>5) + 0..15
    out = torch.zeros(64, 32, dtype=torch.uint8)
    for l in range(64):
        h = l >> 5
        out[l, :16] = rk[l & 31, 16 * h:16 * h + 16]
        out[l, 16:] = rk[l & 31, 32 + 16 * h:32 + 16 * h + 16]
    return out
one = torch.full((64,), 127)
for name, f in (("contig32", img_contig), ("inter16", img_inter16)):
    D = run(f(None, Ab), f(None, Bb.t().contiguous()), one, one)
    print(name, "unit scales: max err vs A@B", float((D - ref_noscale).abs().max()), "ref max", float(ref_noscale.abs().max()))
# scales: hypothesis = byte `opsel` of the lane's scale dword is the E8M0 of (row l&31, k-block l>>5)
def scale_img(s, byte):
    out = torch.zeros(64, dtype=torch.int64)
    for l in range(64):
        out[l] = int(s[l & 31, l >> 5]) << (8 * byte)
    return out
for (opa, opb) in ((0, 0), (1, 1), (2, 3)):
    D = run(img_contig(None, Ab), img_contig(None, Bb.t().contiguous()), scale_img(sA, opa) | 0, scale_img(sB, opb), opa, opb)
    print(f"scales in byte {opa}/{opb} (contig32): max err vs MX reference", float((D - ref).abs().max()), "ref max", float(ref.abs().max()))

print("---- diagnostics")
a_img, b_img = img_contig(None, Ab), img_contig(None, Bb.t().contiguous())
base = run(a_img, b_img, one, one)
def ratio(D):
    r = D / base
    return r
D = run(a_img, b_img, torch.full((64,), 128), one)
print("sa=128 all lanes byte0: ratio min/max", float(ratio(D).min()), float(ratio(D).max()))
D = run(a_img, b_img, torch.full((64,), 128 << 8), one)
print("sa=128 in byte1, opsel 0: ratio", float(ratio(D).min()), float(ratio(D).max()))
D = run(a_img, b_img, torch.full((64,), 128 << 8), one, 1, 1)
print("sa=128 in byte1, opsel 1 (sb=127 in byte 0 -> byte1 of sb is 0!): ratio", float(ratio(D).min()), float(ratio(D).max()))
sa = torch.full((64,), 127); sa[3] = 129
D = run(a_img, b_img, sa, one)
r = ratio(D)
print("sa lane 3 = 129: rows with ratio != 1:", [(i, float(r[i].mean())) for i in range(32) if abs(float(r[i].mean()) - 1) > 1e-3])
sa = torch.full((64,), 127); sa[35] = 129
D = run(a_img, b_img, sa, one)
# expected if (row 3, k-block 1): D[3] = A[3,:32]B[:32] + 4 A[3,32:]B[32:]
exp = Af[3, :32] @ Bf[:32] + 4 * (Af[3, 32:] @ Bf[32:])
print("sa lane 35 = 129: row 3 err vs (kblock1 x4):", float((D[3] - exp).abs().max()), " other rows changed:", [i for i in range(32) if i != 3 and float((D[i] - base[i]).abs().max()) > 1e-3])
sb = torch.full((64,), 127); sb[37] = 130
D = run(a_img, b_img, one, sb)
exp = Af[:, :32] @ Bf[:32, 5] + 8 * (Af[:, 32:] @ Bf[32:, 5])
print("sb lane 37 = 130: col 5 err vs (kblock1 x8):", float((D[:, 5] - exp).abs().max()), " other cols changed:", [j for j in range(32) if j != 5 and float((D[:, j] - base[:, j]).abs().max()) > 1e-3])

print("---- which scale lanes apply to which (lane-half, 16-byte register chunk) of A / B")
for h in range(2):
    for c in range(2):
        am = torch.zeros_like(a_img)
        am[32 * h:32 * h + 32, 16 * c:16 * c + 16] = a_img[32 * h:32 * h + 32, 16 * c:16 * c + 16]
        b0 = run(am, b_img, one, one)
        sa = torch.full((64,), 127); sa[:32] = 129
        lo = run(am, b_img, sa, one)
        sa = torch.full((64,), 127); sa[32:] = 129
        hi = run(am, b_img, sa, one)
        sb = torch.full((64,), 127); sb[:32] = 130
        blo = run(am, b_img, one, sb)
        sb = torch.full((64,), 127); sb[32:] = 130
        bhi = run(am, b_img, one, sb)
        f = lambda D: float((D.abs().sum() / b0.abs().sum()))
        print(f"A chunk (half {h}, bytes {16*c}..{16*c+15}): sa lanes<32 x{f(lo):.2f}, sa lanes>=32 x{f(hi):.2f};  sb lanes<32 x{f(blo):.2f}, sb lanes>=32 x{f(bhi):.2f}")
