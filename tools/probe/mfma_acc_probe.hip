// mfma_acc_probe.hip - does the ORDER in which a wave visits its accumulators and operand fragments change what the board sustains under its power
// cap?  Pure v_mfma_f32_16x16x32_bf16 loops, random operands, no memory traffic, 8 waves per CU (2 per SIMD), 24 accumulator blocks per wave
// (the 6 x 4 grid of the GEMM's 96 x 64 wave tile), two fragment sets P / Q as in a K step of gemm_sp_kernel.
//   MODE 0: row-major per half (rounds 1-5 of the GEMM): P-half over the 24 blocks, then Q-half
//   MODE 1: row-major serpentine per half (the GEMM's order since round 5)
//   MODE 2: row by row, both halves of a row before the next row: (r, c0..c3, P) then (r, c3..c0, Q) - every block's two MFMAs of a K step are
//           at most 7 instructions apart, the last / first one back to back on the same accumulator
//   MODE 3: row by row, block by block: (r, c, P), (r, c, Q) back to back on the same accumulator for every block
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_acc_probe mfma_acc_probe.hip ; run: ./mfma_acc_probe [iters] [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ bf16x8 rnd(unsigned s0, int k) {
    union { unsigned u[4]; bf16x8 v; } x;
    for (int i = 0; i < 4; ++i) x.u[i] = ((s0 * (2654435761u + i + 17 * k) + 977u * k) & 0x807f807fu) | 0x3f003f00u;
    return x.v;
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* sink, int iters) {
    const unsigned s0 = 1u + 1103515245u * (unsigned)(blockIdx.x * blockDim.x + threadIdx.x + 1);
    bf16x8 fa[2][6], fw[2][4];
    for (int h = 0; h < 2; ++h) {
        for (int r = 0; r < 6; ++r) fa[h][r] = rnd(s0, h * 16 + r);
        for (int c = 0; c < 4; ++c) fw[h][c] = rnd(s0 ^ 0x9e3779b9u, h * 16 + 8 + c);
    }
    f32x4 acc[6][4] = {};
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int c = (MODE == 1 && (r & 1)) ? 3 - cc : cc;
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[h][c], fa[h][r], acc[r][c], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);   // (pins the order, as in gemm_sp_kernel)
                    }
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0][c], fa[0][r], acc[r][c], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int c = 3; c >= 0; --c) { acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1][c], fa[1][r], acc[r][c], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
            }
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0][c], fa[0][r], acc[r][c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1][c], fa[1][r], acc[r][c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else if constexpr (MODE == 7) {   // pairs, the K halves in alternating order (P Q | Q P | ...): the A-row operand changes every second instruction
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int h0 = c & 1;
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[h0][c], fa[h0][r], acc[r][c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1 - h0][c], fa[1 - h0][r], acc[r][c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {   // MODE 6: chains of four (a K step of 128 would give a block four MFMAs): half the blocks per loop iteration, same instruction count
            if (it & 1) {
#pragma unroll
                for (int r = 3; r < 6; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int k = 0; k < 4; ++k) { acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[k & 1][c], fa[k & 1][r], acc[r][c], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int k = 0; k < 4; ++k) { acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[k & 1][c], fa[k & 1][r], acc[r][c], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        // (keeps the loop from being re-ordered or hoisted across iterations; costs nothing)
        asm volatile("" ::: "memory");
    }
    float t = 0.f;
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 4; ++c) for (int e = 0; e < 4; ++e) t += acc[r][c][e];
    if (t == 12345.678f) sink[0] = t;
}

// MODE 4 / 5: the same wave tile on v_mfma_f32_32x32x16_bf16 (3 x 2 blocks of 32 x 32, two K = 16 sub-steps per half): 4 = sub-step outermost inside a half
// (the order of the GEMM's 32x32 form until round 4), 5 = the two sub-steps of a block back to back on its accumulator
template <int MODE>
__global__ __launch_bounds__(512) void probe32(float* sink, int iters) {
    const unsigned s0 = 1u + 1103515245u * (unsigned)(blockIdx.x * blockDim.x + threadIdx.x + 1);
    bf16x8 fa[2][2][3], fw[2][2][2];   // [half][k sub-step][block]
    for (int h = 0; h < 2; ++h)
        for (int k = 0; k < 2; ++k) {
            for (int r = 0; r < 3; ++r) fa[h][k][r] = rnd(s0, h * 16 + k * 8 + r);
            for (int c = 0; c < 2; ++c) fw[h][k][c] = rnd(s0 ^ 0x9e3779b9u, h * 16 + k * 8 + 4 + c);
        }
    f32x16 acc[3][2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (MODE == 4) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) { acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[h][k][c], fa[h][k][r], acc[r][c], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int k = 0; k < 2; ++k) { acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[h][k][c], fa[h][k][r], acc[r][c], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        asm volatile("" ::: "memory");
    }
    float t = 0.f;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) t += acc[r][c][e];
    if (t == 12345.678f) sink[0] = t;
}

static int g_threads = 512;
template <int MODE>
static double run(float* sink, int cus, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    if constexpr (MODE == 4 || MODE == 5) hipLaunchKernelGGL(probe32<MODE>, dim3(cus), dim3(g_threads), 0, 0, sink, iters);
    else hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(g_threads), 0, 0, sink, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return (double)cus * (g_threads / 64.0) * iters * 48.0 * 16384.0 / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 60000;   // ~0.25 s per launch
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    g_threads = argc > 3 ? atoi(argv[3]) : 512;   // 256: one wave per SIMD (nobody to cover a dependent pair)
    printf("%d threads per workgroup (%d waves per SIMD)\n", g_threads, g_threads / 256);
    float* sink; (void)hipMalloc(&sink, 64);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    run<0>(sink, cus, iters);   // warm-up: the board reaches its power-capped state
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        const double t[8] = {run<0>(sink, cus, iters), run<1>(sink, cus, iters), run<2>(sink, cus, iters), run<3>(sink, cus, iters), run<4>(sink, cus, iters), run<5>(sink, cus, iters), run<6>(sink, cus, iters), run<7>(sink, cus, iters)};
        printf("round %d: 16x16x32 row-major %7.1f  serpentine %7.1f  row: P then Q reversed %7.1f  block: P, Q back to back %7.1f | 32x32x16 sub-step outermost %7.1f  sub-steps back to back %7.1f | 16x16x32 chains of four %7.1f  pairs, K halves alternating %7.1f TFLOP/s\n",
               r, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
        for (int m = 0; m < 8; ++m) sum[m] += t[m];
    }
    printf("mean   :");
    for (int m = 0; m < 8; ++m) printf("  mode %d %7.1f (%+.2f %%)", m, sum[m] / rounds, 100.0 * (sum[m] / sum[0] - 1));
    printf("\n");
    return 0;
}
