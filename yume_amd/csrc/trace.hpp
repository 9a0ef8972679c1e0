// trace.hpp — per-workgroup time stamps for EXPERIMENT builds only (tools/build_variant.sh <tag> <file.hip> -DYUME_TRACE): where inside a launch
// the time goes (dispatch gaps, prologue, K loop, epilogue; which XCD / CU ran what). The product build compiles none of this.
//   per workgroup 8 words: slots 0..6 = s_memrealtime (100 MHz) at the points the kernel chooses (0 = entry, 1 = K loop done, 2 = end,
//   3..6 = finer points of the kernel under study); word 7: XCC_ID << 32 | HW_ID
#pragma once
#ifdef YUME_TRACE
#include <hip/hip_runtime.h>
#define YUME_TRACE_MAX 32768
// SINGLE-TRANSLATION-UNIT by contract: the buffer and its reader are defined HERE, so exactly one .hip file of a library may be compiled
// with -DYUME_TRACE (tools/build_variant.sh rebuilds one file per variant and takes every other object from the product build). A second
// traced file fails at link time on the duplicate yume_debug_trace_read symbol — deliberately: each unit would otherwise get its own
// buffer and the reader would return only one of them.
__device__ unsigned long long g_yume_trace[YUME_TRACE_MAX * 8];
__device__ __forceinline__ void trace_stamp(int slot) {
    if (threadIdx.x == 0 && blockIdx.x < YUME_TRACE_MAX) {
        g_yume_trace[blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memrealtime();
        if (slot == 0)
            g_yume_trace[blockIdx.x * 8 + 7] =
                ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}
extern "C" __attribute__((visibility("default"))) int yume_debug_trace_read(void* dst, long long bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_yume_trace), (size_t)bytes);
}
#define TRACE_STAMP(s) trace_stamp(s)
#else
#define TRACE_STAMP(s) ((void)0)
#endif
