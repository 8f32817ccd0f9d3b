// mfma_shape_probe.hip - which bf16 MFMA shape does this board sustain best under its power cap?  Pure MFMA loops with random operands,
// no memory traffic, 8 waves per CU (2 per SIMD): v_mfma_f32_32x32x16_bf16 (16 accumulator registers, 32 K flop, 32 cycles) against
// v_mfma_f32_16x16x32_bf16 (4 accumulator registers, 16 K flop, 16 cycles): the same FLOP per cycle, other register traffic per FLOP
// (accumulators: 32 vs 16 register accesses per 32 K flop; A / B operands: 8 vs 16).  Round 4: the big GEMMs are energy bound.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void operands(bf16x8* a, bf16x8* b, int n) {
    unsigned s0 = 1u + 1103515245u * (unsigned)(blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (int k = 0; k < n; ++k) {
        union { unsigned u[4]; bf16x8 v; } x, y;
        for (int i = 0; i < 4; ++i) {
            x.u[i] = ((s0 * (2654435761u + i + 17 * k)) & 0x807f807fu) | 0x3f003f00u;
            y.u[i] = ((s0 * (40503u + 7 * i + 29 * k) + i) & 0x807f807fu) | 0x3f003f00u;
        }
        a[k] = x.v; b[k] = y.v;
    }
}
// MODE 0: 32x32x16, 4 independent accumulator tiles, one operand pair (the round-2 probe)
// MODE 1: 16x16x32, 8 independent accumulator tiles (the same 32 K flop per 32 cycles per pair of instructions), one operand pair
// MODE 2 / 3: the same with 4 distinct operand pairs cycling (closer to a GEMM's register traffic)
template <int MODE>
__global__ __launch_bounds__(512) void probe(float* sink, int iters, unsigned long long* clk) {
    bf16x8 a[4], b[4];
    operands(a, b, 4);
    float acc = 0.f;
    __syncthreads();
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    if constexpr (MODE == 0 || MODE == 2) {
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int it = 0; it < iters; ++it) {
            constexpr int S = MODE == 2 ? 1 : 0;
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1 * S], b[1 * S], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * S], b[2 * S], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3 * S], b[3 * S], c3, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) acc += c0[r] + c1[r] + c2[r] + c3[r];
    } else {
        f32x4 c[8] = {};
        for (int it = 0; it < iters; ++it) {
            constexpr int S = MODE == 3 ? 1 : 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(k & 3) * S], b[(k & 3) * S], c[k], 0, 0, 0);
        }
        for (int k = 0; k < 8; ++k) for (int r = 0; r < 4; ++r) acc += c[k][r];
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (acc == 12345.678f) sink[0] = acc;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int MODE>
static void run(const char* name, float* sink, int cus, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* clk; hipMalloc(&clk, 16);
    hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(512), 0, 0, sink, 2000, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(512), 0, 0, sink, iters, clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)cus * 8.0 * iters * 4.0 * 32768.0;   // per iteration: 4 x 32 K flop (MODE 0 / 2) = 8 x 16 K flop (MODE 1 / 3)
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost); hipFree(clk);
    // shader cycles per 32 K flop per wave (one 32x32x16, or two 16x16x32): 64 = the pipe's rate with two waves per SIMD sharing it
    printf("%-58s %8.3f ms  %7.0f TFLOP/s  %.3f GHz  %.1f cycles per 32 Kflop per wave\n", name, ms, flop / (ms * 1e-3) / 1e12,
           (double)hc[0] / ((double)hc[1] * 10.0), (double)hc[0] / (4.0 * iters));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 150000;
    float* sink; hipMalloc(&sink, 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("32x32x16, 4 accumulator tiles, 1 operand pair", sink, cus, iters);
        run<1>("16x16x32, 8 accumulator tiles, 1 operand pair", sink, cus, iters);
        run<2>("32x32x16, 4 accumulator tiles, 4 operand pairs", sink, cus, iters);
        run<3>("16x16x32, 8 accumulator tiles, 4 operand pairs", sink, cus, iters);
    }
    return 0;
}
