// norm.hip — HBM-bound row kernels of the DiT block:
//   * fused LayerNorm(no affine) + modulate -> bf16 / fp32 / 3-way bf16 split
//   * RMSNorm over the full hidden dim (+ interleaved-pair 3D RoPE), in place on bf16 q|k
// One 256-thread workgroup per token row; every global access is a 16-byte vector.
// (r5, measured and removed: ONE WAVE per row — 64 lanes x 12 vectors, statistics by two wave reductions, no LDS and no barrier, four
// independent rows per workgroup, 16 rows in flight per CU. Slower at every shape of the block: adaLN 37.6 vs 31.3 us, RMSNorm+RoPE 43.3 vs
// 42.0 us at the 5B width, 251 vs 219 / 288 vs 242 us at the 14B width (profiles/r5_norm_wave_ab.json; git show 307463c:yume_amd/csrc/norm.hip
// has the kernels): a wave's twelve 1 KiB loads go out one behind the other, while a block's 256 threads fetch a row in three 4 KiB pieces.)
// Roofline: HBM. Algorithmic bytes per token: adaln 4C read + 2C write; rmsnorm_rope 2*nparts*C r+w.
#include "common.hpp"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float block_sum(float v, float* red /*[8]*/) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6;
    __syncthreads();  // protect `red` against the previous use
    if ((threadIdx.x & 63) == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// MAXV = max float4 vectors per thread (C <= NT*4*MAXV)
// RMS = true: T5LayerNorm (wan/modules/t5.py:53-67): y = x * rsqrt(mean(x^2) + eps) * w — no mean subtraction, no shift
template <int MAXV, bool RMS = false>
__global__ __launch_bounds__(NT) void adaln_kernel(const float* __restrict__ x, int64_t ldx, int C, float eps,
                                                   const float* __restrict__ mul, const float* __restrict__ add,
                                                   int64_t tab_stride, const int32_t* __restrict__ row_idx,
                                                   float add_one, void* __restrict__ out, int64_t ldo,
                                                   int out_kind) {
    __shared__ float red[8];
    const int64_t t = blockIdx.x;
    const float* xr = x + t * ldx;
    const int nvec = C >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = threadIdx.x + i * NT;
        if (vi < nvec) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * vi);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // the modulation rows do not depend on the statistics: their loads go out now, under the two block reductions (narrow rows only:
    // at MAXV = 5 / 8 the extra live registers halve the blocks per CU)
    constexpr bool HOIST = MAXV <= 3;
    const int64_t row = row_idx ? (int64_t)row_idx[t] : 0;
    const float* mr = mul + row * tab_stride;
    const float* ar = RMS ? mr : add + row * tab_stride;
    f32x4 m4v[MAXV], a4v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = threadIdx.x + i * NT;
        if (HOIST && vi < nvec) {
            m4v[i] = *reinterpret_cast<const f32x4*>(mr + 4 * vi);
            a4v[i] = RMS ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(ar + 4 * vi);
        }
    }
    const float mean = RMS ? 0.f : block_sum(s, red) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = threadIdx.x + i * NT;
        if (vi < nvec) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                q += d * d;
            }
        }
    }
    const float var = block_sum(q, red) / (float)C;
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = threadIdx.x + i * NT;
        if (vi < nvec) {
            const f32x4 m4 = HOIST ? m4v[i] : *reinterpret_cast<const f32x4*>(mr + 4 * vi);
            const f32x4 a4 = HOIST ? a4v[i] : RMS ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(ar + 4 * vi);
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * (m4[j] + add_one) + a4[j];
            if (out_kind == 1) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + t * ldo + 4 * vi) = y;
            } else {
                unsigned short* ob = reinterpret_cast<unsigned short*>(out) + t * ldo;
                u32x2 hi;
                hi[0] = pack_bf16x2(y[0], y[1]);
                hi[1] = pack_bf16x2(y[2], y[3]);
                *reinterpret_cast<u32x2*>(ob + 4 * vi) = hi;
                if (out_kind == 2) {
                    // [hi | hi | lo]: lo = bf16(y - float(hi)); A'=[hi,hi,lo] x W'=[Whi,Wlo,Whi] ~ fp32 product
                    *reinterpret_cast<u32x2*>(ob + C + 4 * vi) = hi;
                    float r[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[j] = y[j] - bf16_to_f32(f32_to_bf16(y[j]));
                    u32x2 lo;
                    lo[0] = pack_bf16x2(r[0], r[1]);
                    lo[1] = pack_bf16x2(r[2], r[3]);
                    *reinterpret_cast<u32x2*>(ob + 2 * C + 4 * vi) = lo;
                }
            }
        }
    }
}

// (r6) TWO rows per workgroup for the block's own shape class (C <= 3072, bf16 out): the loads of both rows and of their modulation rows are
// in flight before the first reduction, and one pair of barriers serves both rows' statistics — per row the same arithmetic in the same order
// as adaln_kernel<3> (bit-identical results). Measured in the step: adaLN 33.6 -> see profiles/r6_glue_two_rows_ab.log.
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red /*[16]*/) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int wid = threadIdx.x >> 6;
    __syncthreads();  // protect `red` against the previous use
    if ((threadIdx.x & 63) == 0) { red[wid] = a; red[8 + wid] = b; }
    __syncthreads();
    float ta = 0.f, tb = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) { ta += red[i]; tb += red[8 + i]; }
    a = ta;
    b = tb;
}

__global__ __launch_bounds__(NT) void adaln2_kernel(const float* __restrict__ x, int64_t ldx, int T, int C, float eps,
                                                    const float* __restrict__ mul, const float* __restrict__ add,
                                                    int64_t tab_stride, const int32_t* __restrict__ row_idx,
                                                    float add_one, unsigned short* __restrict__ out, int64_t ldo) {
    constexpr int MAXV = 3;
    __shared__ float red[16];
    const int64_t t0 = 2 * (int64_t)blockIdx.x;
    const bool two = t0 + 1 < T;                                   // (uniform) the last block of an odd T holds one row
    const int64_t tr[2] = {t0, two ? t0 + 1 : t0};
    const int nvec = C >> 2;
    f32x4 v[2][MAXV], m4v[2][MAXV], a4v[2][MAXV];
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = threadIdx.x + i * NT;
            if (vi < nvec) {
                v[r][i] = *reinterpret_cast<const f32x4*>(x + tr[r] * ldx + 4 * vi);
            } else {
                v[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    // (the two rows of a block nearly always carry the same modulation row — the rows of a timestep are contiguous tokens —: one set of loads then)
    const int64_t mrow[2] = {row_idx ? (int64_t)row_idx[tr[0]] : 0, row_idx ? (int64_t)row_idx[tr[1]] : 0};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r == 1 && mrow[1] == mrow[0]) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                m4v[1][i] = m4v[0][i];
                a4v[1][i] = a4v[0][i];
            }
            break;
        }
        const float* mr = mul + mrow[r] * tab_stride;
        const float* ar = add + mrow[r] * tab_stride;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = threadIdx.x + i * NT;
            if (vi < nvec) {
                m4v[r][i] = *reinterpret_cast<const f32x4*>(mr + 4 * vi);
                a4v[r][i] = *reinterpret_cast<const f32x4*>(ar + 4 * vi);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < MAXV; ++i) s[r] += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
    block_sum2(s[0], s[1], red);
    const float mean[2] = {s[0] / (float)C, s[1] / (float)C};
    float q[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = threadIdx.x + i * NT;
            if (vi < nvec) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[r][i][j] - mean[r];
                    q[r] += d * d;
                }
            }
        }
    block_sum2(q[0], q[1], red);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r == 1 && !two) break;
        const float rstd = rsqrtf(q[r] / (float)C + eps);
        unsigned short* ob = out + tr[r] * ldo;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = threadIdx.x + i * NT;
            if (vi < nvec) {
                f32x4 y;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = (v[r][i][j] - mean[r]) * rstd * (m4v[r][i][j] + add_one) + a4v[r][i][j];
                u32x2 hi;
                hi[0] = pack_bf16x2(y[0], y[1]);
                hi[1] = pack_bf16x2(y[2], y[3]);
                *reinterpret_cast<u32x2*>(ob + 4 * vi) = hi;
            }
        }
    }
}

// RMSNorm(+RoPE) in place. Each thread owns NV 16-byte vectors (8 bf16) of the row:
// vector index vi = tid + i*NT, i < NV; part(vi) = vi / (C/8) is wave-uniform since (C/8)%64==0.
template <int NV>
__global__ __launch_bounds__(NT) void rmsnorm_rope_kernel(unsigned short* __restrict__ buf, int64_t ld, int C,
                                                          int nparts, const float* __restrict__ w, float eps,
                                                          const float* __restrict__ rope /*[T,64,2] or null*/, int wperiod) {
    __shared__ float red[8];
    const int64_t t = blockIdx.x;
    if (wperiod > 1) w += (int64_t)(t % wperiod) * nparts * C;   // row t uses weight row t % wperiod (one launch for all layers' K)
    unsigned short* row = buf + t * ld;
    const int vpp = C >> 3;  // vectors per part
    const int nvec = vpp * nparts;
    u16x8 v[NV];
    float ss[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = threadIdx.x + i * NT;
        if (vi < nvec) {
            v[i] = *reinterpret_cast<const u16x8*>(row + 8 * vi);
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = bf16_to_f32(v[i][j]);
                a += f * f;
            }
            if (vi < vpp) ss[0] += a; else ss[1] += a;
        }
    }
    const float s0 = block_sum(ss[0], red);
    float s1 = 0.f;
    if (nparts > 1) s1 = block_sum(ss[1], red);
    // eps < 0: the rows are NOT normalised (the reference's nn.Identity in place of WanRMSNorm, qk_norm=False): y = x * w, then RoPE
    const float r0 = eps < 0.f ? 1.f : rsqrtf(s0 / (float)C + eps);
    const float r1 = eps < 0.f ? 1.f : rsqrtf(s1 / (float)C + eps);
    const float* rp = rope ? rope + t * 128 : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = threadIdx.x + i * NT;
        if (vi < nvec) {
            const bool p1 = vi >= vpp;
            const float r = p1 ? r1 : r0;
            const int c0 = 8 * vi;  // column in [0, nparts*C) ; weight index identical
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + c0);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(w + c0 + 4);
            float y[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                y[j] = bf16_to_f32(v[i][j]) * r * w0[j];
                y[4 + j] = bf16_to_f32(v[i][4 + j]) * r * w1[j];
            }
            if (rp) {
                // element e within the head: (c0 % 128) .. +7  -> complex pairs (c0%128)/2 .. +3
                const int pr = (c0 & 127) >> 1;
                const f32x4 cs0 = *reinterpret_cast<const f32x4*>(rp + 2 * pr);      // cos0 sin0 cos1 sin1
                const f32x4 cs1 = *reinterpret_cast<const f32x4*>(rp + 2 * pr + 4);  // cos2 sin2 cos3 sin3
                const float cs[8] = {cs0[0], cs0[1], cs0[2], cs0[3], cs1[0], cs1[1], cs1[2], cs1[3]};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float a = y[2 * p], b = y[2 * p + 1];
                    const float c = cs[2 * p], s = cs[2 * p + 1];
                    y[2 * p] = a * c - b * s;
                    y[2 * p + 1] = a * s + b * c;
                }
            }
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(y[2 * j], y[2 * j + 1]);
            *reinterpret_cast<u32x4*>(row + 8 * vi) = o;
        }
    }
}

// two rows per workgroup (r6; see adaln2_kernel): per row the arithmetic of rmsnorm_rope_kernel in the same order
template <int NV>
__global__ __launch_bounds__(NT) void rmsnorm_rope2_kernel(unsigned short* __restrict__ buf, int64_t ld, int T, int C, int nparts,
                                                           const float* __restrict__ w, float eps, const float* __restrict__ rope) {
    __shared__ float red[32];
    const int64_t t0 = 2 * (int64_t)blockIdx.x;
    const bool two = t0 + 1 < T;
    const int64_t tr[2] = {t0, two ? t0 + 1 : t0};
    const int vpp = C >> 3;
    const int nvec = vpp * nparts;
    u16x8 v[2][NV];
    float ss[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = threadIdx.x + i * NT;
            if (vi < nvec) v[r][i] = *reinterpret_cast<const u16x8*>(buf + tr[r] * ld + 8 * vi);
        }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = threadIdx.x + i * NT;
            if (vi < nvec) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = bf16_to_f32(v[r][i][j]);
                    a += f * f;
                }
                if (vi < vpp) ss[r][0] += a; else ss[r][1] += a;
            }
        }
    {   // the four sums in one pair of barriers (each: wave sum, then the waves' partials in wave order, as block_sum)
        float p[4] = {wave_sum(ss[0][0]), wave_sum(ss[0][1]), wave_sum(ss[1][0]), wave_sum(ss[1][1])};
        const int wid = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red[8 * k + wid] = p[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float tt = 0.f;
#pragma unroll
            for (int i = 0; i < NT / 64; ++i) tt += red[8 * k + i];
            p[k] = tt;
        }
        ss[0][0] = p[0]; ss[0][1] = p[1]; ss[1][0] = p[2]; ss[1][1] = p[3];
    }
    // The auxiliary loads are hoisted (r6, second half): a thread's NV vectors of a row all sit at the same offset inside their head
    // ((8 * NT) % 128 == 0), so its RoPE pairs are ONE 32-byte read per row, not one per vector; the weights of vector i serve both rows.
    // As written before (4 auxiliary 16-byte loads per 16-byte vector of data) the kernel moved five times its data through the CU's
    // texture path. Same arithmetic in the same order: same bits.
    static_assert((8 * NT) % 128 == 0, "a thread's vectors share their offset inside the head");
    float rs[2][2];
    f32x4 cs0[2], cs1[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        rs[r][0] = eps < 0.f ? 1.f : rsqrtf(ss[r][0] / (float)C + eps);
        rs[r][1] = eps < 0.f ? 1.f : rsqrtf((nparts > 1 ? ss[r][1] : 0.f) / (float)C + eps);
        if (rope) {
            const float* rp = rope + tr[r] * 128 + 2 * (((8 * threadIdx.x) & 127) >> 1);
            cs0[r] = *reinterpret_cast<const f32x4*>(rp);
            cs1[r] = *reinterpret_cast<const f32x4*>(rp + 4);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = threadIdx.x + i * NT;
        if (vi >= nvec) continue;
        const int c0 = 8 * vi;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + c0);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(w + c0 + 4);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r == 1 && !two) break;
            const float rr = vi >= vpp ? rs[r][1] : rs[r][0];
            float y[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                y[j] = bf16_to_f32(v[r][i][j]) * rr * w0[j];
                y[4 + j] = bf16_to_f32(v[r][i][4 + j]) * rr * w1[j];
            }
            if (rope) {
                const float cs[8] = {cs0[r][0], cs0[r][1], cs0[r][2], cs0[r][3], cs1[r][0], cs1[r][1], cs1[r][2], cs1[r][3]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a = y[2 * q], b = y[2 * q + 1];
                    const float c = cs[2 * q], sn = cs[2 * q + 1];
                    y[2 * q] = a * c - b * sn;
                    y[2 * q + 1] = a * sn + b * c;
                }
            }
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(y[2 * j], y[2 * j + 1]);
            *reinterpret_cast<u32x4*>(buf + tr[r] * ld + 8 * vi) = o;
        }
    }
}

}  // namespace

extern "C" int yume_adaln_modulate(const float* x, int64_t ldx, int64_t T, int64_t C, float eps, const float* mul,
                                   const float* add, int64_t tab_stride, const int32_t* row_idx, int add_one,
                                   void* out, int64_t ldo, int out_kind, void* stream) {
    YUME_REQUIRE(x && mul && add && out, "adaln_modulate: NULL pointer");
    YUME_REQUIRE(T >= 0 && C > 0 && (C % 8) == 0 && C <= 8192, "adaln_modulate: C=%lld must be a multiple of 8 and <= 8192", (long long)C);
    YUME_REQUIRE(out_kind >= 0 && out_kind <= 2, "adaln_modulate: out_kind %d", out_kind);
    YUME_REQUIRE((ldx % 4) == 0 && (ldo % 4) == 0 && (tab_stride % 4) == 0, "adaln_modulate: strides must be multiples of 4");
    if (T == 0) return YUME_OK;
    hipStream_t st = (hipStream_t)stream;
    const float one = add_one ? 1.f : 0.f;
    dim3 grid((unsigned)T), block(NT);
    static const bool two_rows = [] { const char* v = getenv("YUME_NORM_TWO_ROWS"); return !v || atoi(v) != 0; }();
    if (two_rows && C <= NT * 4 * 3 && out_kind == 0 && T >= 1024 && T < (1ll << 31))
        hipLaunchKernelGGL(adaln2_kernel, dim3((unsigned)((T + 1) / 2)), block, 0, st, x, ldx, (int)T, (int)C, eps, mul, add, tab_stride, row_idx, one,
                           (unsigned short*)out, ldo);
    else if (C <= NT * 4 * 3)
        hipLaunchKernelGGL(adaln_kernel<3>, grid, block, 0, st, x, ldx, (int)C, eps, mul, add, tab_stride, row_idx, one, out, ldo, out_kind);
    else if (C <= NT * 4 * 5)
        hipLaunchKernelGGL(adaln_kernel<5>, grid, block, 0, st, x, ldx, (int)C, eps, mul, add, tab_stride, row_idx, one, out, ldo, out_kind);
    else
        hipLaunchKernelGGL(adaln_kernel<8>, grid, block, 0, st, x, ldx, (int)C, eps, mul, add, tab_stride, row_idx, one, out, ldo, out_kind);
    YUME_CHECK_LAUNCH("adaln_modulate");
    return YUME_OK;
}

extern "C" int yume_rmsnorm_f32(const float* x, int64_t ldx, int64_t T, int64_t C, float eps, const float* w, void* out,
                                int64_t ldo, void* stream) {
    YUME_REQUIRE(x && w && out, "rmsnorm_f32: NULL pointer");
    YUME_REQUIRE(T >= 0 && C > 0 && (C % 8) == 0 && C <= 8192, "rmsnorm_f32: C=%lld must be a multiple of 8 and <= 8192", (long long)C);
    YUME_REQUIRE((ldx % 4) == 0 && (ldo % 4) == 0, "rmsnorm_f32: strides must be multiples of 4");
    if (T == 0) return YUME_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)T), block(NT);
    if (C <= NT * 4 * 3)
        hipLaunchKernelGGL((adaln_kernel<3, true>), grid, block, 0, st, x, ldx, (int)C, eps, w, w, (int64_t)0, (const int32_t*)nullptr, 0.f, out, ldo, 0);
    else if (C <= NT * 4 * 5)
        hipLaunchKernelGGL((adaln_kernel<5, true>), grid, block, 0, st, x, ldx, (int)C, eps, w, w, (int64_t)0, (const int32_t*)nullptr, 0.f, out, ldo, 0);
    else
        hipLaunchKernelGGL((adaln_kernel<8, true>), grid, block, 0, st, x, ldx, (int)C, eps, w, w, (int64_t)0, (const int32_t*)nullptr, 0.f, out, ldo, 0);
    YUME_CHECK_LAUNCH("rmsnorm_f32");
    return YUME_OK;
}

static int rmsnorm_rope_impl(void* buf, int64_t ld, int64_t T, int64_t C, int nparts, const float* w, float eps,
                             const float* rope, int64_t D, int wperiod, void* stream);

extern "C" int yume_rmsnorm_rope(void* buf, int64_t ld, int64_t T, int64_t C, int nparts, const float* w, float eps,
                                 const float* rope, int64_t D, void* stream) {
    return rmsnorm_rope_impl(buf, ld, T, C, nparts, w, eps, rope, D, 1, stream);
}

// RMSNorm over C of T rows where row t takes weight row (t % wperiod): the K projections of all `wperiod` blocks' cross-attention
// normalised in one launch (the buffer [tokens, wperiod*C] viewed as [tokens*wperiod, C])
extern "C" int yume_rmsnorm_rows_periodic(void* buf, int64_t ld, int64_t T, int64_t C, const float* w, int64_t wperiod, float eps,
                                          void* stream) {
    YUME_REQUIRE(wperiod >= 1 && wperiod <= 1024, "rmsnorm_rows_periodic: wperiod %lld", (long long)wperiod);
    return rmsnorm_rope_impl(buf, ld, T, C, 1, w, eps, nullptr, 128, (int)wperiod, stream);
}

static int rmsnorm_rope_impl(void* buf, int64_t ld, int64_t T, int64_t C, int nparts, const float* w, float eps,
                             const float* rope, int64_t D, int wperiod, void* stream) {
    YUME_REQUIRE(buf && w, "rmsnorm_rope: NULL pointer");
    YUME_REQUIRE(nparts == 1 || nparts == 2, "rmsnorm_rope: nparts must be 1 or 2");
    YUME_REQUIRE(C > 0 && (C % 512) == 0 && C <= 8192, "rmsnorm_rope: C=%lld must be a multiple of 512 and <= 8192", (long long)C);
    YUME_REQUIRE(rope == nullptr || D == 128, "rmsnorm_rope: RoPE needs head_dim 128");
    YUME_REQUIRE((ld % 8) == 0, "rmsnorm_rope: ld must be a multiple of 8");
    if (T == 0) return YUME_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned short* b = reinterpret_cast<unsigned short*>(buf);
    const int64_t nvec = (C / 8) * nparts;
    dim3 grid((unsigned)T), block(NT);
    static const bool two_rows = [] { const char* v = getenv("YUME_NORM_TWO_ROWS"); return !v || atoi(v) != 0; }();
    if (two_rows && wperiod == 1 && nvec <= NT * 3 && T >= 1024 && T < (1ll << 31)) {
        const dim3 g2((unsigned)((T + 1) / 2));
        if (nvec <= NT * 2) hipLaunchKernelGGL(rmsnorm_rope2_kernel<2>, g2, block, 0, st, b, ld, (int)T, (int)C, nparts, w, eps, rope);
        else hipLaunchKernelGGL(rmsnorm_rope2_kernel<3>, g2, block, 0, st, b, ld, (int)T, (int)C, nparts, w, eps, rope);
        YUME_CHECK_LAUNCH("rmsnorm_rope");
        return YUME_OK;
    }
    if (nvec <= NT * 2)
        hipLaunchKernelGGL(rmsnorm_rope_kernel<2>, grid, block, 0, st, b, ld, (int)C, nparts, w, eps, rope, wperiod);
    else if (nvec <= NT * 3)
        hipLaunchKernelGGL(rmsnorm_rope_kernel<3>, grid, block, 0, st, b, ld, (int)C, nparts, w, eps, rope, wperiod);
    else if (nvec <= NT * 5)
        hipLaunchKernelGGL(rmsnorm_rope_kernel<5>, grid, block, 0, st, b, ld, (int)C, nparts, w, eps, rope, wperiod);
    else
        hipLaunchKernelGGL(rmsnorm_rope_kernel<8>, grid, block, 0, st, b, ld, (int)C, nparts, w, eps, rope, wperiod);
    YUME_CHECK_LAUNCH("rmsnorm_rope");
    return YUME_OK;
}
