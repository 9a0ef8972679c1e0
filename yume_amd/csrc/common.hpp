// common.hpp — shared device/host helpers for libyume_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/yume_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#define WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------
void yume_set_error(const char* fmt, ...);

#define YUME_REQUIRE(cond, ...)                      \
    do {                                             \
        if (!(cond)) {                               \
            yume_set_error(__VA_ARGS__);             \
            return YUME_EINVAL;                      \
        }                                            \
    } while (0)

#define YUME_CHECK_LAUNCH(name)                                                       \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            yume_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return YUME_ELAUNCH;                                                      \
        }                                                                             \
    } while (0)

// ---- bf16 <-> f32 (device) ------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
    return __uint_as_float(((unsigned int)h) << 16);
}
// round-to-nearest-even in hardware: both lower to v_cvt_pk_bf16_f32 on gfx950
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    return (unsigned short)(pack_bf16x2(f, 0.f) & 0xffffu);
}

// ---- wave / block reductions --------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// tanh-approximated GELU exactly as torch.nn.GELU(approximate='tanh'):
//   0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715 x^3)))
// With tanh(u) = 1 - 2/(exp(2u)+1) the expression is x / (1 + exp(-2u)) (no cancellation in either tail): one v_exp_f32 (exp2 of the
// pre-scaled argument), one v_rcp_f32 (1 ulp, no IEEE division sequence) and five full-rate ops per element — the GEMM epilogue
// evaluates it 128 times per lane with the matrix pipe idle. Saturates cleanly: exp2 -> 0 gives x, exp2 -> inf gives x * 0.
__device__ __forceinline__ float gelu_tanh(float x) {
    const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;   // -2 * sqrt(2/pi) * log2(e)
    const float c1 = c0 * 0.044715f;
    const float x2 = x * x;
    const float e = __builtin_amdgcn_exp2f(x * (c1 * x2 + c0));            // exp(-2u)
    return x * __builtin_amdgcn_rcpf(e + 1.0f);
}
// two elements at a time on the packed fp32 ALU (v_pk_mul / v_pk_fma / v_pk_add: one issue slot for both) — the same operations in the same
// order as gelu_tanh, so the same bits; for epilogues that run with the matrix pipe idle and are bound by VALU issue
__device__ __forceinline__ f32x2_t gelu_tanh2(f32x2_t x) {
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = k0 * 0.044715f;
    const f32x2_t c0 = {k0, k0}, c1 = {k1, k1}, one = {1.0f, 1.0f};
    const f32x2_t x2 = x * x;
    const f32x2_t u = x * (c1 * x2 + c0);
    const f32x2_t e = {__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};
    const f32x2_t d = e + one;
    const f32x2_t r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return x * r;
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
