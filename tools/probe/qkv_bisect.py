"""Time the QKV (mode 4, head-norm + RoPE) GEMM of the metric shape with a given build of the library (ctypes, raw)."""
import ctypes as C, sys, torch
lib = C.CDLL(sys.argv[1])
dev = torch.device("cuda:0")
M, N, K = 6000, 4096, 2048
A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
wq = torch.ones(128, device=dev); wk = torch.ones(128, device=dev)
f = lib.ace355_gemm_bf16_headnorm
f.restype = C.c_int
f.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p] * 2 + [C.c_float, C.c_int, C.c_int, C.c_float, C.c_void_p]
for _ in range(6):
    rc = f(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, 2048, 3072, wq.data_ptr(), wk.data_ptr(), 1e-6, 1, 375, 1e6, None)
    assert rc == 0, rc
torch.cuda.synchronize()
# plain bf16 store GEMM for comparison (mode 0)
g = lib.ace355_gemm_bf16
g.restype = C.c_int
g.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p] * 2
for _ in range(6):
    assert g(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, 1, None, None) == 0
torch.cuda.synchronize()
