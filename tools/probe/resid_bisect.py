"""Time the residual (mode 2) and SwiGLU (mode 3) GEMMs of the metric shapes with a given build of the library (ctypes, raw)."""
import ctypes as C, sys, torch
lib = C.CDLL(sys.argv[1])
dev = torch.device("cuda:0")
r = lib.ace355_gemm_bf16_residual
r.restype = C.c_int
r.argtypes = [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p] * 2 + [C.c_int] * 2 + [C.c_void_p, C.c_int, C.c_void_p]
fz = lib.ace355_gemm_bf16_fused
fz.restype = C.c_int
fz.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p] * 2 + [C.c_int] * 2 + [C.c_void_p]
for (M, N, K) in [(6000, 2048, 2048), (6000, 2048, 6144), (3000, 2048, 2048)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    H = torch.zeros(M, N, device=dev); g1 = torch.randn(N, device=dev); g2 = torch.randn(64, N, device=dev)
    for _ in range(6):
        assert r(A.data_ptr(), W.data_ptr(), H.data_ptr(), M, N, K, g1.data_ptr(), g2.data_ptr(), N, 375, None, 0, None) == 0
M, N, K = 6000, 12288, 2048
A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    assert fz(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, 1, None, None, 0, 0, None) == 0
torch.cuda.synchronize()
