// misc.hip — library plumbing + the small HBM-bound helpers around the DiT block stack:
// sinusoidal timestep embedding, small-M fp32 linear (time MLP on R distinct timesteps),
// patch gather (im2col for stride==kernel Conv3d), unpatchify, cast/pad, transpose.
#include "common.hpp"
#include "counters.hpp"
#include <string.h>
#include <atomic>

// ---- error plumbing -----------------------------------------------------------------------
static thread_local char g_err[512] = "";

void yume_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* yume_last_error(void) { return g_err; }
extern "C" int yume_abi_version(void) { return YUME_ABI_VERSION; }
extern "C" const char* yume_target_arch(void) { return "gfx950"; }

// ---- caller-owned ticket-counter workspace (counters.hpp) -------------------------------------------
namespace yume_counters {
constexpr int MAXDEV = 64;
static std::atomic<int*> g_base[MAXDEV];
static std::atomic<unsigned> g_next[MAXDEV];
int* next_set() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    int* b = g_base[dev].load(std::memory_order_acquire);
    return b ? b + SET_INTS * (g_next[dev].fetch_add(1u, std::memory_order_relaxed) % SETS) : nullptr;
}
}  // namespace yume_counters

extern "C" int64_t yume_counter_workspace_bytes(void) { return (int64_t)yume_counters::SETS * yume_counters::SET_INTS * (int64_t)sizeof(int); }

extern "C" int yume_counter_workspace_init(void* ptr, int64_t bytes, void* stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= yume_counters::MAXDEV) {
        yume_set_error("counter_workspace_init: no current device");
        return YUME_ELAUNCH;
    }
    if (ptr == nullptr) {                                   // unregister: ticketed launches fall back to their static schedules
        yume_counters::g_base[dev].store(nullptr, std::memory_order_release);
        return YUME_OK;
    }
    YUME_REQUIRE(bytes >= yume_counter_workspace_bytes(), "counter_workspace_init: %lld bytes given, %lld needed", (long long)bytes,
                 (long long)yume_counter_workspace_bytes());
    YUME_REQUIRE((reinterpret_cast<uintptr_t>(ptr) & 63) == 0, "counter_workspace_init: the buffer must be 64-byte aligned");
    if (hipMemsetAsync(ptr, 0, (size_t)yume_counter_workspace_bytes(), static_cast<hipStream_t>(stream)) != hipSuccess) {
        yume_set_error("counter_workspace_init: memset failed");
        return YUME_ELAUNCH;
    }
    yume_counters::g_base[dev].store(static_cast<int*>(ptr), std::memory_order_release);
    return YUME_OK;
}

namespace {

// ---- sinusoidal embedding: out[r, :] = [cos(t*w_i) | sin(t*w_i)], w_i = 10000^(-i/half), fp64 math ----
__global__ void sinus_kernel(const double* __restrict__ t, const int32_t* __restrict__ t_index, int R, int dim,
                             float* __restrict__ out) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * half) return;
    const int r = i / half, c = i % half;
    const double pos = t[t_index ? t_index[r] : r];
    const double w = pow(10000.0, -((double)c) / (double)half);
    const double a = pos * w;
    out[(int64_t)r * dim + c] = (float)cos(a);
    out[(int64_t)r * dim + half + c] = (float)sin(a);
}

// ---- small-M linear: one wave per output feature n, all R rows at once; W streamed once ----
template <bool WBF16, int RMAX>
__global__ __launch_bounds__(256) void linear_smallm_kernel(const float* __restrict__ in, int R, int K,
                                                            const void* __restrict__ Wv, const float* __restrict__ bias,
                                                            int N, int in_act, int out_act,
                                                            const float* __restrict__ add_table, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[r] = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        float w[8];
        if (WBF16) {
            const u16x8 wv = *reinterpret_cast<const u16x8*>(reinterpret_cast<const unsigned short*>(Wv) + (int64_t)n * K + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = bf16_to_f32(wv[j]);
        } else {
            const float* wp = reinterpret_cast<const float*>(Wv) + (int64_t)n * K + k;
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { w[j] = w0[j]; w[4 + j] = w1[j]; }
        }
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            if (r < R) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(in + (int64_t)r * K + k);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(in + (int64_t)r * K + k + 4);
                float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xv = in_act ? silu(x[j]) : x[j];
                    acc[r] = fmaf(xv, w[j], acc[r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
        const float b = (bias ? bias[n] : 0.f);
        const float a = (add_table ? add_table[n] : 0.f);
        for (int r = 0; r < R && r < RMAX; ++r) {
            float v = acc[r] + b;
            if (out_act) v = silu(v);
            out[(int64_t)r * N + n] = v + a;
        }
    }
}

// ---- patch gather: out[(f,hp,wp), (c,dh,dw)] = x[c, f0+f, hp*kh+dh, wp*kw+dw] (0 outside) ----
template <bool INBF16>
__global__ __launch_bounds__(256) void patch_gather_kernel(const void* __restrict__ xv, int Cin, int F, int H, int W,
                                                           int f0, int nf, int kh, int kw, int Hp, int Wp,
                                                           unsigned short* __restrict__ out, int Kp) {
    const int64_t tok = blockIdx.x;  // (f, hp, wp)
    const int wp = (int)(tok % Wp);
    const int hp = (int)((tok / Wp) % Hp);
    const int f = (int)(tok / ((int64_t)Wp * Hp));
    const int Kv = Cin * kh * kw;
    unsigned short* orow = out + tok * Kp;
    for (int col = threadIdx.x; col < Kp; col += blockDim.x) {
        unsigned short val = 0;
        if (col < Kv) {
            const int dw = col % kw;
            const int dh = (col / kw) % kh;
            const int c = col / (kw * kh);
            const int hh = hp * kh + dh, ww = wp * kw + dw;
            if (hh < H && ww < W) {
                const int64_t idx = (((int64_t)c * F + (f0 + f)) * H + hh) * W + ww;
                if (INBF16) val = reinterpret_cast<const unsigned short*>(xv)[idx];
                else val = f32_to_bf16(reinterpret_cast<const float*>(xv)[idx]);
            }
        }
        orow[col] = val;
    }
}

// ---- unpatchify: in[(f,hp,wp), (p,q,c)] -> out[c, f, hp*ph+p, wp*pw+q] ----
__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ in, int64_t ldi, int Fr, int Hp, int Wp,
                                                         int ph, int pw, int Cout, float* __restrict__ out) {
    const int64_t total = (int64_t)Cout * Fr * Hp * ph * Wp * pw;
    const int Wo = Wp * pw, Ho = Hp * ph;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % Wo);
        const int h = (int)((i / Wo) % Ho);
        const int f = (int)((i / ((int64_t)Wo * Ho)) % Fr);
        const int c = (int)(i / ((int64_t)Wo * Ho * Fr));
        const int hp = h / ph, p = h % ph, wq = w / pw, q = w % pw;
        const int64_t tok = ((int64_t)f * Hp + hp) * Wp + wq;
        out[i] = in[tok * ldi + ((int64_t)p * pw + q) * Cout + c];
    }
}

__global__ __launch_bounds__(256) void modtab_kernel(const float* __restrict__ tab, const float* __restrict__ e0, int64_t B,
                                                     int64_t R, int64_t W4, float* __restrict__ out) {
    const int64_t total = B * R * W4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t w = i % W4, r = (i / W4) % R, b = i / (W4 * R);
        const f32x4 a = *reinterpret_cast<const f32x4*>(tab + (b * W4 + w) * 4);
        const f32x4 e = *reinterpret_cast<const f32x4*>(e0 + (r * W4 + w) * 4);
        *reinterpret_cast<f32x4*>(out + i * 4) = a + e;
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, int64_t ldi, int64_t rows_valid,
                                                        int64_t rows, int64_t cols4, unsigned short* __restrict__ out,
                                                        int64_t ldo) {
    const int64_t total = rows * cols4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols4, c = (i % cols4) * 4;
        u32x2 o;
        o[0] = 0u; o[1] = 0u;
        if (r < rows_valid) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(in + r * ldi + c);
            o[0] = pack_bf16x2(v[0], v[1]);
            o[1] = pack_bf16x2(v[2], v[3]);
        }
        *reinterpret_cast<u32x2*>(out + r * ldo + c) = o;
    }
}

// 32x32 tiles through LDS (padded): coalesced on both sides
template <bool INBF16>
__global__ __launch_bounds__(256) void transpose_kernel(const void* __restrict__ in, int64_t ldi, int64_t rows, int64_t cols,
                                                        unsigned short* __restrict__ out, int64_t ldo) {
    __shared__ unsigned short tile[32][33];
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + ty + 8 * i, c = c0 + tx;
        unsigned short v = 0;
        if (r < rows && c < cols) {
            if (INBF16) v = reinterpret_cast<const unsigned short*>(in)[r * ldi + c];
            else v = f32_to_bf16(reinterpret_cast<const float*>(in)[r * ldi + c]);
        }
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < rows && c < cols) out[c * ldo + r] = tile[tx][ty + 8 * i];
    }
}

}  // namespace

extern "C" int yume_sinusoidal_embed(const double* t, const int32_t* t_index, int64_t R, int64_t dim, float* out,
                                     void* stream) {
    YUME_REQUIRE(t && out, "sinusoidal_embed: NULL pointer");
    YUME_REQUIRE(R > 0 && dim > 0 && (dim % 2) == 0, "sinusoidal_embed: bad shape R=%lld dim=%lld", (long long)R, (long long)dim);
    const int64_t n = R * (dim / 2);
    hipLaunchKernelGGL(sinus_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t, t_index,
                       (int)R, (int)dim, out);
    YUME_CHECK_LAUNCH("sinusoidal_embed");
    return YUME_OK;
}

extern "C" int yume_modulation_table(const float* tab, const float* e0, int64_t B, int64_t R, int64_t W, float* out,
                                     void* stream) {
    YUME_REQUIRE(tab && e0 && out, "modulation_table: NULL pointer");
    YUME_REQUIRE(B > 0 && R > 0 && W > 0 && (W % 4) == 0, "modulation_table: bad shape");
    const int64_t total = B * R * (W / 4);
    int64_t nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(modtab_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, tab, e0, B, R, W / 4, out);
    YUME_CHECK_LAUNCH("modulation_table");
    return YUME_OK;
}

extern "C" int yume_linear_smallm_f32(const float* in, int64_t R, int64_t K, const void* W, int w_bf16,
                                      const float* bias, int64_t N, int in_act, int out_act, const float* add_table,
                                      float* out, void* stream) {
    YUME_REQUIRE(in && W && out, "linear_smallm_f32: NULL pointer");
    YUME_REQUIRE(R > 0 && R <= 8, "linear_smallm_f32: R=%lld must be in 1..8", (long long)R);
    YUME_REQUIRE(K > 0 && (K % 8) == 0 && N > 0, "linear_smallm_f32: K=%lld must be a multiple of 8", (long long)K);
    dim3 grid((unsigned)((N + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (w_bf16)
        hipLaunchKernelGGL((linear_smallm_kernel<true, 8>), grid, block, 0, st, in, (int)R, (int)K, W, bias, (int)N, in_act, out_act, add_table, out);
    else
        hipLaunchKernelGGL((linear_smallm_kernel<false, 8>), grid, block, 0, st, in, (int)R, (int)K, W, bias, (int)N, in_act, out_act, add_table, out);
    YUME_CHECK_LAUNCH("linear_smallm_f32");
    return YUME_OK;
}

extern "C" int yume_patch_gather(const void* x, int in_bf16, int64_t Cin, int64_t F, int64_t H, int64_t W, int64_t f0,
                                 int64_t nf, int64_t kh, int64_t kw, void* out, int64_t Kp, void* stream) {
    YUME_REQUIRE(x && out, "patch_gather: NULL pointer");
    YUME_REQUIRE(Cin > 0 && F > 0 && H > 0 && W > 0 && kh > 0 && kw > 0, "patch_gather: bad shape");
    YUME_REQUIRE(f0 >= 0 && nf > 0 && f0 + nf <= F, "patch_gather: frame range [%lld, %lld) outside F=%lld", (long long)f0, (long long)(f0 + nf), (long long)F);
    YUME_REQUIRE(Kp >= Cin * kh * kw, "patch_gather: Kp=%lld < Cin*kh*kw", (long long)Kp);
    const int64_t Hp = (H + kh - 1) / kh, Wp = (W + kw - 1) / kw;
    const int64_t ntok = nf * Hp * Wp;
    YUME_REQUIRE(ntok < (1ll << 31), "patch_gather: too many tokens");
    dim3 grid((unsigned)ntok), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (in_bf16)
        hipLaunchKernelGGL(patch_gather_kernel<true>, grid, block, 0, st, x, (int)Cin, (int)F, (int)H, (int)W, (int)f0, (int)nf, (int)kh, (int)kw, (int)Hp, (int)Wp, (unsigned short*)out, (int)Kp);
    else
        hipLaunchKernelGGL(patch_gather_kernel<false>, grid, block, 0, st, x, (int)Cin, (int)F, (int)H, (int)W, (int)f0, (int)nf, (int)kh, (int)kw, (int)Hp, (int)Wp, (unsigned short*)out, (int)Kp);
    YUME_CHECK_LAUNCH("patch_gather");
    return YUME_OK;
}

extern "C" int yume_unpatchify(const float* in, int64_t ldi, int64_t Fr, int64_t Hp, int64_t Wp, int64_t ph, int64_t pw,
                               int64_t Cout, float* out, void* stream) {
    YUME_REQUIRE(in && out, "unpatchify: NULL pointer");
    YUME_REQUIRE(Fr > 0 && Hp > 0 && Wp > 0 && ph > 0 && pw > 0 && Cout > 0, "unpatchify: bad shape");
    YUME_REQUIRE(ldi >= ph * pw * Cout, "unpatchify: ldi too small");
    const int64_t total = Cout * Fr * Hp * ph * Wp * pw;
    int64_t nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, ldi, (int)Fr, (int)Hp, (int)Wp, (int)ph, (int)pw, (int)Cout, out);
    YUME_CHECK_LAUNCH("unpatchify");
    return YUME_OK;
}

extern "C" int yume_cast_bf16(const float* in, int64_t ldi, int64_t rows_valid, int64_t rows, int64_t cols, void* out,
                              int64_t ldo, void* stream) {
    YUME_REQUIRE(in && out, "cast_bf16: NULL pointer");
    YUME_REQUIRE(rows > 0 && cols > 0 && (cols % 4) == 0 && (ldi % 4) == 0 && (ldo % 4) == 0, "cast_bf16: cols/ldi/ldo must be multiples of 4");
    YUME_REQUIRE(rows_valid >= 0 && rows_valid <= rows, "cast_bf16: rows_valid out of range");
    const int64_t total = rows * (cols / 4);
    int64_t nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, ldi, rows_valid, rows, cols / 4, (unsigned short*)out, ldo);
    YUME_CHECK_LAUNCH("cast_bf16");
    return YUME_OK;
}

extern "C" int yume_transpose_bf16(const void* in, int in_bf16, int64_t ldi, int64_t rows, int64_t cols, void* out,
                                   int64_t ldo, void* stream) {
    YUME_REQUIRE(in && out, "transpose_bf16: NULL pointer");
    YUME_REQUIRE(rows > 0 && cols > 0 && ldi >= cols && ldo >= rows, "transpose_bf16: bad shape");
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)), block(256);
    YUME_REQUIRE(grid.y < 65536, "transpose_bf16: too many rows");
    hipStream_t st = (hipStream_t)stream;
    if (in_bf16)
        hipLaunchKernelGGL(transpose_kernel<true>, grid, block, 0, st, in, ldi, rows, cols, (unsigned short*)out, ldo);
    else
        hipLaunchKernelGGL(transpose_kernel<false>, grid, block, 0, st, in, ldi, rows, cols, (unsigned short*)out, ldo);
    YUME_CHECK_LAUNCH("transpose_bf16");
    return YUME_OK;
}

// ---- per-box calibration (bench.py): what THIS chip sustains under a pure matrix-pipe load ------------------------------------------
// `workgroups` x 4 waves (one per SIMD, as the product kernels), each issuing iters x 16 v_mfma_f32_32x32x16_bf16 on two independent
// accumulator sets and random operands, and nothing else: 2 * 32 * 32 * 16 flop per instruction, 32 clocks of one SIMD's matrix pipe each. The caller times
// the launch with events on `stream`; ticks[wg] = {s_memtime delta, s_memrealtime delta (100 MHz)} of the workgroup's first wave.
// Identical MI355X boxes differ by several per cent in the clock their power management sustains under such a load: the
// bench line carries this figure so that its rates can be compared across boxes.
namespace {
__global__ __launch_bounds__(256) void calibrate_mfma_kernel(long long iters, unsigned long long* ticks, float* sink) {
    // RANDOM operands, a different pair for each of the 8 MFMA slots of the loop body: the energy of a matrix instruction follows the
    // bits that toggle — with constant operands the same launch runs at 2.31 GHz / 2.42 PFLOP/s (r5, first build), no power limit in sight.
    f32x16 a0 = {}, a1 = {};
    bf16x8_t x[8], y[8];
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        u32x4 wx, wy;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // two bf16 per word: sign and 7 mantissa bits random, exponent field 125 or 127 (|v| in [0.25, 0.5) or [1, 2)) — finite sums whatever the count
            h = h * 1664525u + 1013904223u;
            wx[i] = (h & 0x80ff80ffu) | 0x3e803e80u | ((h >> 3) & 0x01800180u);
            h = h * 1664525u + 1013904223u;
            wy[i] = (h & 0x80ff80ffu) | 0x3e803e80u | ((h >> 5) & 0x01800180u);
        }
        x[k] = __builtin_bit_cast(bf16x8_t, wx);
        y[k] = __builtin_bit_cast(bf16x8_t, wy);
    }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (long long it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[j], y[j], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[j], x[(j + 3) & 7], a1, 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    if (s == 12345.678f) sink[0] = s;                 // keeps the accumulators alive; (practically) never true
    if (threadIdx.x == 0 && ticks) {
        ticks[2 * blockIdx.x] = c1 - c0;
        ticks[2 * blockIdx.x + 1] = r1 - r0;
    }
}
}  // namespace

extern "C" int yume_calibrate_mfma(int64_t iters, int64_t workgroups, void* ticks, void* sink, void* stream) {
    YUME_REQUIRE(iters > 0 && workgroups > 0 && workgroups <= 65535, "calibrate_mfma: iters / workgroups out of range");
    YUME_REQUIRE(sink != nullptr, "calibrate_mfma: sink must point at 4 bytes of device memory");
    hipLaunchKernelGGL(calibrate_mfma_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, (long long)iters,
                       (unsigned long long*)ticks, (float*)sink);
    YUME_CHECK_LAUNCH("calibrate_mfma");
    return YUME_OK;
}
