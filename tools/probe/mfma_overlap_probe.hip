// mfma_overlap_probe.hip - may the destination of v_mfma_f32_16x16x32_bf16 overlap one of its SOURCE operands (D == A or D == B, C elsewhere)?
// LLVM treats 4-register MFMA destinations as free to overlap srcA / srcB (no early-clobber), and hipcc used that freedom in one instantiation
// of gemm_sp_kernel (round 6: the condition encoder's gate|up launch was not reproducible).  Here every lane computes the same MFMA chain in three
// register forms - reference (D == C), D == B, D == A - in several pipeline contexts (PRE independent MFMAs issued back to back ahead of it,
// POST behind it), REPS times; the forms must agree bit for bit and every repeat must agree with the first.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_overlap_probe mfma_overlap_probe.hip ; run: ./mfma_overlap_probe [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ bf16x8 rnd(unsigned s0, int k) {
    union { unsigned u[4]; bf16x8 v; } x;
    for (int i = 0; i < 4; ++i) x.u[i] = ((s0 * (2654435761u + i + 17 * k) + 977u * k) & 0x807f807fu) | 0x3f003f00u;
    return x.v;
}

// FORM 0: D == C (in place); 1: D == B (the B operand's registers receive the result, C is read from elsewhere); 2: D == A
template <int FORM, int PRE, int POST>
__global__ __launch_bounds__(256) void probe(float* out) {
    const unsigned s0 = 1u + 1103515245u * (unsigned)(blockIdx.x * blockDim.x + threadIdx.x + 1);
    bf16x8 a = rnd(s0, 1), b = rnd(s0 ^ 0x9e3779b9u, 2);
    bf16x8 pa[4], pb[4];
    f32x4 pacc[4], c = {0.25f, -0.5f, 1.0f, 2.0f};
    for (int i = 0; i < 4; ++i) { pa[i] = rnd(s0, 10 + i); pb[i] = rnd(s0, 20 + i); pacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x4 d;
    // busy pipe: PRE independent MFMAs right ahead of the one under test (asm: one block, nothing the compiler can slip in between)
    if constexpr (FORM == 0) {
        d = c;
        asm volatile(
            ".if %c12 > 0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %9, %1\n\t.endif\n\t"
            ".if %c12 > 1\n\tv_mfma_f32_16x16x32_bf16 %2, %6, %10, %2\n\t.endif\n\t"
            "v_mfma_f32_16x16x32_bf16 %0, %7, %8, %0\n\t"
            ".if %c13 > 0\n\tv_mfma_f32_16x16x32_bf16 %3, %5, %10, %3\n\t.endif\n\t"
            ".if %c13 > 1\n\tv_mfma_f32_16x16x32_bf16 %4, %6, %9, %4\n\t.endif\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7"
            : "+v"(d), "+v"(pacc[0]), "+v"(pacc[1]), "+v"(pacc[2]), "+v"(pacc[3])
            : "v"(pa[0]), "v"(pa[1]), "v"(a), "v"(b), "v"(pb[0]), "v"(pb[1]), "v"(0), "n"(PRE), "n"(POST));
    } else if constexpr (FORM == 1) {
        f32x4 db = __builtin_bit_cast(f32x4, b);
        asm volatile(
            ".if %c12 > 0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %9, %1\n\t.endif\n\t"
            ".if %c12 > 1\n\tv_mfma_f32_16x16x32_bf16 %2, %6, %10, %2\n\t.endif\n\t"
            "v_mfma_f32_16x16x32_bf16 %0, %7, %0, %8\n\t"
            ".if %c13 > 0\n\tv_mfma_f32_16x16x32_bf16 %3, %5, %10, %3\n\t.endif\n\t"
            ".if %c13 > 1\n\tv_mfma_f32_16x16x32_bf16 %4, %6, %9, %4\n\t.endif\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7"
            : "+v"(db), "+v"(pacc[0]), "+v"(pacc[1]), "+v"(pacc[2]), "+v"(pacc[3])
            : "v"(pa[0]), "v"(pa[1]), "v"(a), "v"(c), "v"(pb[0]), "v"(pb[1]), "v"(0), "n"(PRE), "n"(POST));
        d = db;
    } else {
        f32x4 da = __builtin_bit_cast(f32x4, a);
        asm volatile(
            ".if %c12 > 0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %9, %1\n\t.endif\n\t"
            ".if %c12 > 1\n\tv_mfma_f32_16x16x32_bf16 %2, %6, %10, %2\n\t.endif\n\t"
            "v_mfma_f32_16x16x32_bf16 %0, %0, %7, %8\n\t"
            ".if %c13 > 0\n\tv_mfma_f32_16x16x32_bf16 %3, %5, %10, %3\n\t.endif\n\t"
            ".if %c13 > 1\n\tv_mfma_f32_16x16x32_bf16 %4, %6, %9, %4\n\t.endif\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7"
            : "+v"(da), "+v"(pacc[0]), "+v"(pacc[1]), "+v"(pacc[2]), "+v"(pacc[3])
            : "v"(pa[0]), "v"(pa[1]), "v"(b), "v"(c), "v"(pb[0]), "v"(pb[1]), "v"(0), "n"(PRE), "n"(POST));
        d = da;
    }
    float* o = out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 8;
    for (int e = 0; e < 4; ++e) o[e] = d[e];
    for (int e = 0; e < 4; ++e) o[4 + e] = pacc[0][e] + pacc[1][e] + pacc[2][e] + pacc[3][e];
}

template <int FORM, int PRE, int POST>
static void run(float* dev, float* host, int n) {
    hipLaunchKernelGGL((probe<FORM, PRE, POST>), dim3(1024), dim3(256), 0, 0, dev);
    hipMemcpy(host, dev, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
}

template <int PRE, int POST>
static int context(float* dev, int n, int reps) {
    float *ref = (float*)malloc(n * 4), *x = (float*)malloc(n * 4);
    run<0, PRE, POST>(dev, ref, n);
    long bad[3] = {0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        run<0, PRE, POST>(dev, x, n); for (int i = 0; i < n; ++i) bad[0] += memcmp(&x[i], &ref[i], 4) != 0;
        run<1, PRE, POST>(dev, x, n); for (int i = 0; i < n; ++i) bad[1] += (i % 8 < 4) && memcmp(&x[i], &ref[i], 4) != 0;
        run<2, PRE, POST>(dev, x, n); for (int i = 0; i < n; ++i) bad[2] += (i % 8 < 4) && memcmp(&x[i], &ref[i], 4) != 0;
    }
    printf("PRE %d POST %d: mismatching floats over %d repeats: D==C %ld, D==B %ld, D==A %ld (of %d results each)\n", PRE, POST, reps, bad[0], bad[1], bad[2], n / 2);
    free(ref); free(x);
    return (bad[0] | bad[1] | bad[2]) != 0;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const int n = 1024 * 256 * 8;
    float* dev;
    hipMalloc(&dev, (size_t)n * sizeof(float));
    int rc = 0;
    rc |= context<0, 0>(dev, n, reps);
    rc |= context<1, 0>(dev, n, reps);
    rc |= context<2, 0>(dev, n, reps);
    rc |= context<0, 2>(dev, n, reps);
    rc |= context<2, 2>(dev, n, reps);
    printf(rc ? "OVERLAP FORMS DISAGREE\n" : "all forms agree\n");
    return 0;
}
