// mx_probe.hip - layout probe for v_mfma_scale_f32_32x32x64_f8f6f4 (MX block-scaled fp8, gfx950): the host supplies every lane's
// register image (8 dwords of A, 8 of B, one scale dword each) and gets every lane's 16 accumulators back, so that operand /
// scale layouts can be tested as hypotheses from Python (tools/probe/mx_probe.py).  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int OPA, int OPB>
__global__ void mx_kernel(const int* a, const int* b, const int* sa, const int* sb, float* c) {
    const int l = threadIdx.x;
    i32x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = a[l * 8 + i]; B[i] = b[l * 8 + i]; }
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, OPA, sa[l], OPB, sb[l]);
    for (int i = 0; i < 16; ++i) c[l * 16 + i] = acc[i];
}

extern "C" int mx_probe(const int* a, const int* b, const int* sa, const int* sb, float* c, int opa, int opb) {
    if (opa == 0 && opb == 0) hipLaunchKernelGGL((mx_kernel<0, 0>), dim3(1), dim3(64), 0, 0, a, b, sa, sb, c);
    else if (opa == 1 && opb == 1) hipLaunchKernelGGL((mx_kernel<1, 1>), dim3(1), dim3(64), 0, 0, a, b, sa, sb, c);
    else if (opa == 2 && opb == 3) hipLaunchKernelGGL((mx_kernel<2, 3>), dim3(1), dim3(64), 0, 0, a, b, sa, sb, c);
    else return 1;
    return hipDeviceSynchronize() == hipSuccess ? 0 : 2;
}
