"""Time the MXFP8 GEMMs (test hook) of the metric shapes with a given build of the library (ctypes, raw)."""
import ctypes as C, sys, torch
lib = C.CDLL(sys.argv[1])
dev = torch.device("cuda:0")
f = lib.ace355_gemm_mxfp8
f.restype = C.c_int
f.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p] * 2 + [C.c_int] * 2 + [C.c_void_p]
M = 6000
for (N, K, mode) in [(4096, 2048, 0), (12288, 2048, 3), (2048, 6144, 2), (2048, 2048, 2)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.zeros(M, N, device=dev) if mode == 2 else torch.empty(M, N // 2 if mode == 3 else N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        assert f(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, mode, None, None, 0, 375, None) == 0
torch.cuda.synchronize()
