// Micro-benchmark: can a SIMD overlap one wave's MFMAs with VALU work (a) of ANOTHER wave, (b) of the SAME wave?
// 1 workgroup per CU, 8 waves (2 per SIMD): waves 0..3 and 4..7 share SIMDs 0..3.
//   mode 0: all 8 waves MFMA only            mode 1: all 8 waves VALU (fma) only      mode 2: all 8 waves exp only
//   mode 3: waves 0-3 MFMA, waves 4-7 fma    mode 4: waves 0-3 MFMA, waves 4-7 exp
//   mode 5: every wave: MFMA + 6 fma interleaved per MFMA (same-wave co-issue)
//   mode 6: every wave: MFMA + 3 exp interleaved
//   mode 7: waves 0-3 MFMA only, waves 4-7 idle (reference for one wave per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool second = wave >= 4;
    f32x16 acc0 = {}, acc1 = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = 0.001f * (threadIdx.x + 3 * i);
    const float c1 = 0.999f, c2 = 0.0001f;
    const bool do_mfma = MODE == 0 || MODE == 5 || MODE == 6 || ((MODE == 3 || MODE == 4 || MODE == 7) && !second);
    const bool do_fma = MODE == 1 || (MODE == 3 && second);
    const bool do_exp = MODE == 2 || (MODE == 4 && second);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 5 || MODE == 6) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                if (MODE == 5) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) v[q] = __builtin_fmaf(v[q], c1, c2);
                } else {
#pragma unroll
                    for (int q = 0; q < 3; ++q) v[q] = __builtin_amdgcn_exp2f(v[q] * c2);
                }
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                if (MODE == 5) {
#pragma unroll
                    for (int q = 6; q < 12; ++q) v[q] = __builtin_fmaf(v[q], c1, c2);
                } else {
#pragma unroll
                    for (int q = 3; q < 6; ++q) v[q] = __builtin_amdgcn_exp2f(v[q] * c2);
                }
            }
        } else if (do_mfma) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            }
        } else if (do_fma) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 12; ++q) v[q] = __builtin_fmaf(v[q], c1, c2);
        } else if (do_exp) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 6; ++q) v[q] = __builtin_amdgcn_exp2f(v[q] * c2);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    // per iteration: 16 MFMA (32x32x16) per MFMA-wave; 96 fma or 48 exp per VALU-wave
    const char* names[] = {"all MFMA", "all fma (96/iter)", "all exp (48/iter)", "A: MFMA | B: fma", "A: MFMA | B: exp",
                           "same wave MFMA + 6 fma each", "same wave MFMA + 3 exp each", "A: MFMA | B: idle"};
    float ms[8];
    ms[0] = run<0>(out, iters); ms[1] = run<1>(out, iters); ms[2] = run<2>(out, iters); ms[3] = run<3>(out, iters);
    ms[4] = run<4>(out, iters); ms[5] = run<5>(out, iters); ms[6] = run<6>(out, iters); ms[7] = run<7>(out, iters);
    for (int m = 0; m < 8; ++m) {
        const double ns_per_iter = ms[m] * 1e6 / iters;
        printf("mode %d %-32s %8.3f ms  %7.1f ns/iter  (%.0f clk @2.4GHz)\n", m, names[m], ms[m], ns_per_iter, ns_per_iter * 2.4);
    }
    return 0;
}
