// Micro-benchmark (round 6): what the matrix pipe SUSTAINS under the package power limit with each bf16 MFMA shape, on
// random operands (the energy of an MFMA follows the toggling bits: yume_calibrate_mfma, misc.hip), one wave per SIMD on every CU as in
// the product GEMM / attention kernels, nothing else in the loop.
//   mode 0: v_mfma_f32_32x32x16_bf16, 2 independent accumulators (32 regs)           = the calibration kernel of the library
//   mode 1: v_mfma_f32_16x16x32_bf16, 8 independent accumulators (32 regs)           = the GEMM's instruction, operands re-used like a register-blocked tile
//   mode 2: v_mfma_f32_16x16x32_bf16, 64 accumulators (256 regs), 8 A x 8 B operands = exactly the register-blocked 128 x 128 wave tile of gemm_w4
//   mode 3: v_mfma_f32_32x32x16_bf16, 16 accumulators (256 regs), 4 A x 4 B operands = the same wave tile on the 32x32 shape
// Prints TFLOP/s and the implied matrix-pipe clock (16 / 32 pipe clocks per instruction nominal).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ bf16x8 rnd(unsigned& h) {
    u32x4 w;
    for (int i = 0; i < 4; ++i) {
        h = h * 1664525u + 1013904223u;
        w[i] = (h & 0x80ff80ffu) | 0x3e803e80u | ((h >> 3) & 0x01800180u);
    }
    return __builtin_bit_cast(bf16x8, w);
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* sink, long long iters) {
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    bf16x8 x[8], y[8];
    for (int i = 0; i < 8; ++i) { x[i] = rnd(h); y[i] = rnd(h); }
    float s = 0.f;
    if constexpr (MODE == 0) {
        f32x16 a0 = {}, a1 = {};
        for (long long it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[j], y[j], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[j], x[(j + 3) & 7], a1, 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    } else if constexpr (MODE == 1) {
        f32x4 a[8] = {};
        for (long long it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) a[(j * 4 + q) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[j], y[(j + q) & 7], a[(j * 4 + q) & 7], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1] + a[i][2] + a[i][3];
    } else if constexpr (MODE == 2) {
        f32x4 a[64] = {};
        for (long long it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[i * 8 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[i], y[j], a[i * 8 + j], 0, 0, 0);
        }
        for (int i = 0; i < 64; ++i) s += a[i][0] + a[i][1] + a[i][2] + a[i][3];
    } else {
        f32x16 a[16] = {};
        for (long long it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i + 4 * ks], y[j + 4 * ks], a[i * 4 + j], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i)
            for (int q = 0; q < 16; ++q) s += a[i][q];
    }
    if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
void run(float* sink, int ncu, double secs) {
    const double flop_per_iter = MODE == 0 ? 16.0 * 32768 : MODE == 1 ? 32.0 * 16384 : MODE == 2 ? 64.0 * 16384 : 32.0 * 32768;
    const double clk_per_iter = MODE == 0 ? 16.0 * 32 : MODE == 1 ? 32.0 * 16 : MODE == 2 ? 64.0 * 16 : 32.0 * 32;
    const long long iters = MODE == 0 ? 20000 : MODE == 1 ? 20000 : 10000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(256), 0, 0, sink, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(256), 0, 0, sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms1 = 0; hipEventElapsedTime(&ms1, e0, e1);
    int n = (int)(secs * 1e3 / ms1) + 1;
    for (int i = 0; i < n / 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(256), 0, 0, sink, iters);   // settle
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(256), 0, 0, sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double dt = ms * 1e-3;
    const double tf = (double)ncu * 4 * iters * n * flop_per_iter / dt / 1e12;
    const double ghz = (double)iters * n * clk_per_iter / dt / 1e9;
    printf("{\"mode\": %d, \"tflops\": %.1f, \"implied_pipe_clock_ghz\": %.3f, \"launches\": %d, \"seconds\": %.3f}\n", MODE, tf, ghz, n, dt);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 0.5;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    float* sink; hipMalloc(&sink, 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(sink, pr.multiProcessorCount, secs);
        run<1>(sink, pr.multiProcessorCount, secs);
        run<2>(sink, pr.multiProcessorCount, secs);
        run<3>(sink, pr.multiProcessorCount, secs);
    }
    return 0;
}
