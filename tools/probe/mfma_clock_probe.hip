// mfma_clock_probe.hip - what do clock64() / wall_clock64() / rocm-smi mean on this chip, and what clock does it hold under a
// pure-MFMA load?  Every wave issues ITER x 4 independent v_mfma_f32_32x32x16_bf16 (32 cycles each on its SIMD, no memory
// traffic) and records clock64 / wall_clock64 (100 MHz) deltas.  If clock64 ticks shader cycles, cycles / MFMA = 32 (1 wave per
// SIMD) or 64 (2 waves per SIMD sharing the pipe); GHz = cycles / wall.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512) void probe(const unsigned* __restrict__ seed, unsigned long long* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned s0 = seed[(blockIdx.x * blockDim.x + threadIdx.x) & 1023];
    union { unsigned u[4]; bf16x8 v; } a, b;
    for (int i = 0; i < 4; ++i) { a.u[i] = s0 * (2654435761u + i); b.u[i] = s0 * (40503u + 7 * i) + i; }
    // keep exponents sane: mask to small magnitudes (sign + 7 mantissa bits + exponent around 1.0)
    for (int i = 0; i < 4; ++i) { a.u[i] = (a.u[i] & 0x807f807fu) | 0x3f003f00u; b.u[i] = (b.u[i] & 0x807f807fu) | 0x3f003f00u; }
    if (seed[1023] == 0) { for (int i = 0; i < 4; ++i) a.u[i] = b.u[i] = 0; }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    __syncthreads();
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c3, 0, 0, 0);
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += c0[r] + c1[r] + c2[r] + c3[r];
    if (acc == 12345.678f) sink[0] = acc;
    if (lane == 0) {
        const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = w1 - w0;
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    int cus = 256;
    unsigned* seed; unsigned long long* out; float* sink;
    hipMalloc(&seed, 4096); hipMalloc(&sink, 64);
    std::vector<unsigned> hs(1024);
    for (int zero = 0; zero < 2; ++zero)
        for (int wpc = 4; wpc <= 8; wpc += 4) {
            for (int i = 0; i < 1024; ++i) hs[i] = 1u + 1103515245u * (i + 1);
            hs[1023] = zero ? 0u : 1u;
            hipMemcpy(seed, hs.data(), 4096, hipMemcpyHostToDevice);
            const int waves = cus * wpc;
            hipMalloc(&out, waves * 16);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(probe, dim3(cus), dim3(wpc * 64), 0, 0, seed, out, sink, 1000);  // warm
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(cus), dim3(wpc * 64), 0, 0, seed, out, sink, iters);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(waves * 2);
            hipMemcpy(h.data(), out, waves * 16, hipMemcpyDeviceToHost);
            double cyc = 0, wall = 0;
            for (int w = 0; w < waves; ++w) { cyc += (double)h[2 * w]; wall += (double)h[2 * w + 1]; }
            cyc /= waves; wall /= waves;
            const double mfmas = 4.0 * iters;
            const double tflops = (double)waves * mfmas * 32768.0 / (ms * 1e-3) / 1e12;
            printf("%s data, %d waves/CU: %.1f clock64 ticks per MFMA per wave, clock64/wall = %.3f GHz (wall_clock64 assumed 100 MHz), "
                   "event time %.3f ms vs wall_clock64 %.3f ms, %.0f TFLOP/s => MFMA-issue clock %.3f GHz (32 cycles/MFMA/SIMD)\n",
                   zero ? "zero" : "random", wpc, cyc / mfmas, cyc / (wall * 10.0), ms, wall * 1e-5, tflops,
                   tflops / 2500.0 * 2.4);
            hipFree(out);
        }
    return 0;
}
