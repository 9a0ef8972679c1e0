// attn_fwd.hip — exact-softmax attention forward for head_dim 128, bf16 in / fp32 accumulate / bf16 out (gfx950).
//
// Three kernels compute the same function (yume_attn_fwd picks; tests compare them):
//   attn_fwd_kernel_v4  8 waves, 256 queries per workgroup, one workgroup per CU, software-pipelined across key tiles
//                       (self-attention, Lk >= 1536); see the comment above it
//   attn_fwd_kernel_v2  4 waves, 128 queries, two workgroups per CU, K / V^T tiles by LDS-DMA (cross-attention and the
//                       query rows left over after whole rounds of v4 workgroups)
//   attn_fwd_kernel     as v2 with register-staged tiles (the first version; kept as an independent cross-check)
// Common to all: each wave owns 32 queries and walks the keys in tiles of 64. Everything is computed TRANSPOSED so that
// the softmax row of a query lives in ONE lane (plus its partner lane^32) and never needs LDS or cross-lane shuffles:
//
//   S^T[key, q] = K[key, :] . Q[q, :]        v_mfma_f32_32x32x16_bf16, A = K tile (LDS), B = Q^T (registers)
//                 C layout: col = lane&31 = q, row = key = (r&3) + 8*(r>>2) + 4*(lane>>5)
//   P^T         = exp2(S^T*c - m)            in registers; packed to bf16 it IS the B operand of
//   O^T[d, q]   = V^T[d, key] . P^T[key, q]  A = V^T tile (LDS, K-major image), B = P^T (registers)
//                 with the SAME key<->k-slot assignment on both operands, so no permutation is needed.
//   O^T accumulators keep col = lane&31 = q, so the online-softmax rescale is lane-local too.
//
// K tile  : LDS [64 keys][128 d] bf16, 16-byte chunk c of row r stored at chunk c ^ (r & 15)  (ds_read_b128 conflict-free)
// V^T tile: LDS [128 d][64 keys] bf16; v1: row stride 136 B (ds_read_b64); v2 / v4: 128-byte rows, chunk c of row d at
//           chunk c ^ ((d >> 1) & 7), the swizzle applied on the SOURCE address of the LDS-DMA
// Roofline: MFMA (bf16 dense). Algorithmic work 4*Lq*Lk*128 flop per head.
#include "common.hpp"
#include "attn_args.hpp"
#include "counters.hpp"
#include "trace.hpp"
#include "attn_cross_rk.hpp"

namespace {

constexpr int QW = 32;          // queries per wave
constexpr int NW = 4;           // waves per workgroup
constexpr int QB = QW * NW;     // 128 queries per workgroup
constexpr int KT = 64;          // keys per tile
constexpr int D = 128;
constexpr int K_TILE_BYTES = KT * D * 2;   // 16384
constexpr int VROW = 136;                  // bytes per V^T row in LDS (128 + 8 pad)
constexpr int V_TILE_BYTES = D * VROW;     // 17408
constexpr int BUF_BYTES = K_TILE_BYTES + V_TILE_BYTES;
constexpr float NEG_BIG = -1.0e30f;
constexpr float DEFER_LOG2 = 8.0f;         // deferred-rescale threshold in the log2 domain (P <= 256)

// max / sum across the two 32-lane halves of a wave: one v_permlane32_swap instead of an LDS shuffle
__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct Stage {
    u32x4 k[4];
    u32x4 v[4];
};

__device__ __forceinline__ void stage_load(Stage& s, const AttnArgs& p, int h, int j0, int tid) {
    // K: 64 rows x 256 B: thread -> chunk c = tid&15, rows tid/16 + 16*rr
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        int key = j0 + (tid >> 4) + 16 * rr;
        key = key < p.Lk ? key : p.Lk - 1;
        s.k[rr] = *reinterpret_cast<const u32x4*>(p.K + (int64_t)key * p.ldk + h * D + (tid & 15) * 8);
    }
    // V^T: 128 rows (d) x 128 B (64 keys): thread -> chunk c = tid&7, rows tid/8 + 32*rr
    int kc = j0 + (tid & 7) * 8;                      // first key of this chunk
    const int kmax = (int)p.ldvt - 8;
    const int kload = kc < kmax ? kc : kmax;          // keep the 16-byte load inside the row
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int d = (tid >> 3) + 32 * rr;
        s.v[rr] = *reinterpret_cast<const u32x4*>(p.Vt + (int64_t)(h * D + d) * p.ldvt + kload);
    }
    if (kc + 8 > p.Lk) {
        // keys >= Lk must contribute exactly 0 (their P is 0, but 0 * NaN-bits would poison O): zero them
        const int nvalid = (kload == kc) ? max(p.Lk - kc, 0) : 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned int x = s.v[rr][w];
                if (2 * w >= nvalid) x = 0u;
                else if (2 * w + 1 >= nvalid) x &= 0xffffu;
                s.v[rr][w] = x;
            }
        }
    }
}

__device__ __forceinline__ void stage_store(const Stage& s, char* buf, int tid) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = (tid >> 4) + 16 * rr;
        const int c = tid & 15;
        *reinterpret_cast<u32x4*>(buf + r * 256 + ((c ^ (r & 15)) << 4)) = s.k[rr];
    }
    char* vb = buf + K_TILE_BYTES;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int d = (tid >> 3) + 32 * rr;
        char* dst = vb + d * VROW + (tid & 7) * 16;   // 8-byte aligned only: two b64 writes
        u32x2 lo, hi;
        lo[0] = s.v[rr][0]; lo[1] = s.v[rr][1];
        hi[0] = s.v[rr][2]; hi[1] = s.v[rr][3];
        *reinterpret_cast<u32x2*>(dst) = lo;
        *reinterpret_cast<u32x2*>(dst + 8) = hi;
    }
}

__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int ql = lane & 31;

    // block -> (head, query block): XCD x (= blockIdx % 8) works on heads x, x+8, ... so one head's
    // K/V stay in one XCD's L2 while its query blocks stream through.
    int h, qb;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int hx = (p.H + 7 - xcd) >> 3;            // heads owned by this XCD
        const int per = hx * p.nqb;
        if (idx >= per) return;
        h = xcd + 8 * (idx / p.nqb);
        qb = idx % p.nqb;
    }
    const int q0 = p.q_lo + qb * QB + wave * QW;

    // ---- Q^T fragments (B operand): lane (q = ql, hi) holds Q[q][16*ks + 8*hi .. +7] ----
    bf16x8_t qf[8];
    {
        int q = q0 + ql;
        q = q < p.Lq ? q : p.Lq - 1;
        const unsigned short* qp = p.Q + (int64_t)q * p.ldq + h * D + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * ks);
    }

    f32x16 oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = NEG_BIG;   // running max of the SCALED scores (log2 domain)
    float l_run = 0.f;       // this lane's partial row sum (its 32 of every 64 keys)

    const int nt = (p.Lk + KT - 1) / KT;
    Stage st;
    stage_load(st, p, h, 0, tid);
    stage_store(st, smem, tid);
    if (nt > 1) stage_load(st, p, h, KT, tid);
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        const char* kb = smem + cur * BUF_BYTES;
        const char* vb = kb + K_TILE_BYTES;

        // ---- S^T = K . Q^T : two 32-key blocks x 8 k-steps ----
        f32x16 sacc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[b][r] = 0.f;
            const int row = 32 * b + ql;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int c = 2 * ks + hi;
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kb + row * 256 + ((c ^ (row & 15)) << 4));
                sacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[b], 0, 0, 0);
            }
        }

        // ---- online softmax (lane-local + one exchange with lane^32) ----
        const int j0 = t * KT;
        if (j0 + KT > p.Lk) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = j0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= p.Lk) sacc[b][r] = NEG_BIG;
                }
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[b][r]);
        mx = xhalf_max(mx);                                   // combine with the partner lane (other 32 keys)
        const float m_new = fmaxf(m_run, mx * p.scale_log2);
        // deferred rescale: keep the old reference max while it is within 2^DEFER of the new one for every
        // query of the wave (P <= 2^DEFER then; O/l is invariant to the reference) — the O-wide multiply is
        // skipped on most tiles. The previous tile's P.V is complete here, so O, l and m move together.
        if (!__all(m_new - m_run <= DEFER_LOG2)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[b][r], p.scale_log2, -m_run));
                sacc[b][r] = pv;
                psum += pv;
            }
        l_run += psum;

        // ---- P^T -> bf16 B fragments: k-step s uses block s>>1, regs 8*(s&1) .. +7 ----
        bf16x8_t pf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w[j] = pack_bf16x2(sacc[s >> 1][8 * (s & 1) + 2 * j], sacc[s >> 1][8 * (s & 1) + 2 * j + 1]);
            pf[s] = __builtin_bit_cast(bf16x8_t, w);
        }

        // ---- O^T += V^T . P^T : 4 d-blocks x 4 k-steps ----
        // A fragment of k-step s for lane (d = 32*db + ql, hi): keys base .. base+3 and base+8 .. base+11,
        // base = 32*(s>>1) + 16*(s&1) + 4*hi  (the keys whose P the same lane group supplies in pf[s]).
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const char* vrow = vb + (32 * db + ql) * VROW + 8 * hi;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const char* a = vrow + 64 * (s >> 1) + 32 * (s & 1);
                u32x4 w;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(a);
                const u32x2 hi2 = *reinterpret_cast<const u32x2*>(a + 16);
                w[0] = lo[0]; w[1] = lo[1]; w[2] = hi2[0]; w[3] = hi2[1];
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w), pf[s], oacc[db], 0, 0, 0);
            }
        }

        // ---- stage the next tile, prefetch the one after ----
        if (t + 1 < nt) stage_store(st, smem + (cur ^ 1) * BUF_BYTES, tid);
        __syncthreads();
        if (t + 2 < nt) stage_load(st, p, h, (t + 2) * KT, tid);
        cur ^= 1;
    }

    // ---- epilogue: O[q, d] = O^T[d, q] / l ----
    const float l_tot = xhalf_sum(l_run);
    const float inv = 1.0f / l_tot;
    const int q = q0 + ql;
    if (q < p.Lq) {
        unsigned short* op = p.O + (int64_t)q * p.ldo + h * D + 4 * hi;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = oacc[db][4 * g + 0] * inv, v1 = oacc[db][4 * g + 1] * inv;
                float v2 = oacc[db][4 * g + 2] * inv, v3 = oacc[db][4 * g + 3] * inv;
                u32x2* dst = reinterpret_cast<u32x2*>(op + 32 * db + 8 * g);
                if (p.accumulate) {
                    const u32x2 old = *dst;
                    v0 += bf16_to_f32((unsigned short)(old[0] & 0xffffu));
                    v1 += bf16_to_f32((unsigned short)(old[0] >> 16));
                    v2 += bf16_to_f32((unsigned short)(old[1] & 0xffffu));
                    v3 += bf16_to_f32((unsigned short)(old[1] >> 16));
                }
                u32x2 o;
                o[0] = pack_bf16x2(v0, v1);
                o[1] = pack_bf16x2(v2, v3);
                *dst = o;
            }
    }
}

// =====================================================================================================================
// v2: same math and tile shapes, different data movement.
//   * K and V^T tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass; the
//     bank swizzles are applied on the per-lane SOURCE address (K: chunk ^ (row & 15); V^T: chunk ^ ((row >> 1) & 7))
//     so both tiles are read with conflict-free ds_read_b128.
//   * V^T fragments are 16 contiguous bytes = 8 CONSECUTIVE keys, so the P operand has to hold 8 consecutive keys too:
//     after the exp the packed P words of the two half-waves are exchanged with 8 v_permlane32_swap per tile
//     (lane (q,0) gives its odd 4-key groups, receives the partner's even ones).
//   * only a ragged last tile (Lk % 64 != 0) is register-staged, to zero the keys >= Lk of V^T.
// The DMA of tile t+1 is issued before the compute of tile t and waited for (vmcnt(0)) right before the single
// per-tile barrier, i.e. it has the whole tile of MFMA work to land.
constexpr int V2_BUF = 2 * K_TILE_BYTES;      // K 16 KiB + V^T 16 KiB (128-byte rows, no padding)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// per-thread source pointers of the 8 LDS-DMA pieces of a tile (4 K rounds, 4 V^T rounds), advanced by one tile per use
struct DmaPtrs {
    const unsigned short* k[4];
    const unsigned short* v[4];
};
__device__ __forceinline__ void dma_init(DmaPtrs& dp, const AttnArgs& p, int h, int tid) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = rr * 16 + (tid >> 4);            // K: 64 rows x 16 chunks; full tiles only -> no clamp needed
        dp.k[rr] = p.K + (int64_t)r * p.ldk + h * D + (((tid & 15) ^ (r & 15)) << 3);
        const int d = rr * 32 + (tid >> 3);            // V^T: 128 rows x 8 chunks
        dp.v[rr] = p.Vt + (int64_t)(h * D + d) * p.ldvt + (((tid & 7) ^ ((d >> 1) & 7)) << 3);
    }
}
__device__ __forceinline__ void dma_tile(DmaPtrs& dp, int64_t kstep, char* buf, int wave) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)dp.k[rr], (lds_void_t*)(buf + rr * 4096 + wave * 1024), 16, 0, 0);
        dp.k[rr] += kstep;
    }
    char* vb = buf + K_TILE_BYTES;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)dp.v[rr], (lds_void_t*)(vb + rr * 4096 + wave * 1024), 16, 0, 0);
        dp.v[rr] += KT;
    }
}

// register path for the ragged last tile: same LDS image as dma_tile, keys >= Lk of V^T zeroed (stage_load does it)
__device__ __forceinline__ void stage_store_v2(const Stage& s, char* buf, int tid) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = (tid >> 4) + 16 * rr;
        *reinterpret_cast<u32x4*>(buf + r * 256 + (((tid & 15) ^ (r & 15)) << 4)) = s.k[rr];
    }
    char* vb = buf + K_TILE_BYTES;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int d = (tid >> 3) + 32 * rr;
        *reinterpret_cast<u32x4*>(vb + d * 128 + (((tid & 7) ^ ((d >> 1) & 7)) << 4)) = s.v[rr];
    }
}

template <bool MASK>
__device__ __forceinline__ void tile_body_v2(const char* kb, const AttnArgs& p, const bf16x8_t (&qf)[8],
                                             f32x16 (&oacc)[4], float& m_run, float& l_run, int j0, int ql, int hi,
                                             const int (&koff)[8], const int (&voff)[4]) {
    const char* vb = kb + K_TILE_BYTES;
    f32x16 sacc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[b][r] = 0.f;
    {
        // both 32-key halves per k-step, K fragments two k-steps ahead (3-deep ring, order pinned)
        // row 32b + ql has the same swizzle as row ql: one per-lane offset per k-step + an immediate
        bf16x8_t ka[3], kc[3];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            ka[ks] = *reinterpret_cast<const bf16x8_t*>(kb + koff[ks]);
            kc[ks] = *reinterpret_cast<const bf16x8_t*>(kb + koff[ks] + 32 * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + 2 < 8) {
                ka[(ks + 2) % 3] = *reinterpret_cast<const bf16x8_t*>(kb + koff[ks + 2]);
                kc[(ks + 2) % 3] = *reinterpret_cast<const bf16x8_t*>(kb + koff[ks + 2] + 32 * 256);
                __builtin_amdgcn_sched_barrier(0);
            }
            sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks % 3], qf[ks], sacc[0], 0, 0, 0);
            sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[ks % 3], qf[ks], sacc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // rows >= Lk of the K tile are clamped copies of key Lk-1, so the row max needs no mask; their P is zeroed below
    float mx = sacc[0][0];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[b][r]);
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx * p.scale_log2);
    if (!__all(m_new - m_run <= DEFER_LOG2)) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float pv = __builtin_amdgcn_exp2f(fmaf(sacc[b][r], p.scale_log2, -m_run));
            if (MASK) {
                const int key = j0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * hi;
                pv = key < p.Lk ? pv : 0.f;
            }
            sacc[b][r] = pv;
            psum += pv;
        }
    l_run += psum;

    // ---- P^T fragments of 8 consecutive keys: k-step sg = 2b+e covers keys 16*sg + 8*hi' + j ----
    bf16x8_t pf[4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            u32x4 w;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned ev = pack_bf16x2(sacc[b][8 * e + 2 * i], sacc[b][8 * e + 2 * i + 1]);          // group 2e
                const unsigned od = pack_bf16x2(sacc[b][8 * e + 4 + 2 * i], sacc[b][8 * e + 4 + 2 * i + 1]);  // group 2e+1
                const auto r = __builtin_amdgcn_permlane32_swap(ev, od, false, false);
                w[i] = r[0];        // keys j = 0..3 of this lane's half
                w[2 + i] = r[1];    // keys j = 4..7
            }
            pf[2 * b + e] = __builtin_bit_cast(bf16x8_t, w);
        }

    // ---- O^T += V^T . P^T ----
    // The S accumulators are dead once P is packed: their registers hold a 4-deep ring of V^T fragments, pinned with
    // sched_barriers (left alone, the scheduler emits `ds_read ; s_waitcnt lgkmcnt(0) ; v_mfma` sixteen times).
    // row 32db + ql swizzles like row ql ((d >> 1) & 7 is unchanged by + 32db)
    __builtin_amdgcn_sched_barrier(0);
    bf16x8_t vf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vf[i] = *reinterpret_cast<const bf16x8_t*>(vb + voff[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int db = i >> 2, sg = i & 3;
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[sg], pf[sg], oacc[db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 4 < 16) {
            vf[sg] = *reinterpret_cast<const bf16x8_t*>(vb + voff[sg] + (db + 1) * (32 * 128));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel_v2(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * V2_BUF];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int ql = lane & 31;
    int h, qb;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int hx = (p.H + 7 - xcd) >> 3;
        const int per = hx * p.nqb;
        if (idx >= per) return;
        h = xcd + 8 * (idx / p.nqb);
        qb = idx % p.nqb;
    }
    TRACE_STAMP(0);
    const int q0 = p.q_lo + qb * QB + wave * QW;
    bf16x8_t qf[8];
    {
        int q = q0 + ql;
        q = q < p.Lq ? q : p.Lq - 1;
        const unsigned short* qp = p.Q + (int64_t)q * p.ldq + h * D + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * ks);
    }
    f32x16 oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    int koff[8], voff[4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) koff[ks] = ql * 256 + (((2 * ks + hi) ^ (ql & 15)) << 4);
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) voff[sg] = ql * 128 + (((2 * sg + hi) ^ ((ql >> 1) & 7)) << 4);
    DmaPtrs dp;
    dma_init(dp, p, h, tid);
    const int64_t kstep = (int64_t)KT * p.ldk;

    const int nt = (p.Lk + KT - 1) / KT;
    const bool ragged = (p.Lk % KT) != 0;          // then the LAST tile takes the register path
    const int nsp = gridDim.y, sp = blockIdx.y;    // key-range split (1 split = the whole range)
    const int t0 = (int)((int64_t)nt * sp / nsp), t1 = (int)((int64_t)nt * (sp + 1) / nsp);
    if (t0 > 0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            dp.k[rr] += (int64_t)t0 * kstep;
            dp.v[rr] += (int64_t)t0 * KT;
        }
    }
    if (t0 == nt - 1 && ragged) {
        Stage st;
        stage_load(st, p, h, t0 * KT, tid);
        stage_store_v2(st, smem, tid);
    } else {
        dma_tile(dp, kstep, smem, wave);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TRACE_STAMP(3);

    int cur = 0;
    for (int t = t0; t < t1; ++t) {
        char* kb = smem + cur * V2_BUF;
        char* nb = smem + (cur ^ 1) * V2_BUF;
        const bool has_next = t + 1 < t1;
        const bool next_reg = has_next && ragged && (t + 2 == nt);
        if (has_next && !next_reg) dma_tile(dp, kstep, nb, wave);
        if (t == nt - 1 && ragged)
            tile_body_v2<true>(kb, p, qf, oacc, m_run, l_run, t * KT, ql, hi, koff, voff);
        else
            tile_body_v2<false>(kb, p, qf, oacc, m_run, l_run, t * KT, ql, hi, koff, voff);
        if (next_reg) {
            Stage st;
            stage_load(st, p, h, (t + 1) * KT, tid);
            stage_store_v2(st, nb, tid);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    if (nsp > 1) {            // partial result of this key range
        const float l_part = xhalf_sum(l_run);
        const int q = q0 + ql;
        if (q < p.Lq) {
            const int64_t rows = p.Lq - p.q_lo, r = q - p.q_lo;
            float* po = p.part_o + ((int64_t)sp * rows + r) * ((int64_t)p.H * D) + h * D + 4 * hi;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(po + 32 * db + 8 * g) =
                        f32x4{oacc[db][4 * g + 0], oacc[db][4 * g + 1], oacc[db][4 * g + 2], oacc[db][4 * g + 3]};
            if (hi == 0) {
                float* pm = p.part_ml + (((int64_t)sp * rows + r) * p.H + h) * 2;
                pm[0] = m_run;
                pm[1] = l_part;
            }
        }
        return;
    }

    TRACE_STAMP(1);
    const float l_tot = xhalf_sum(l_run);
    const float inv = 1.0f / l_tot;
    const int q = q0 + ql;
    if (q < p.Lq) {
        unsigned short* op = p.O + (int64_t)q * p.ldo + h * D + 4 * hi;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = oacc[db][4 * g + 0] * inv, v1 = oacc[db][4 * g + 1] * inv;
                float v2 = oacc[db][4 * g + 2] * inv, v3 = oacc[db][4 * g + 3] * inv;
                u32x2* dst = reinterpret_cast<u32x2*>(op + 32 * db + 8 * g);
                if (p.accumulate) {
                    const u32x2 old = *dst;
                    v0 += bf16_to_f32((unsigned short)(old[0] & 0xffffu));
                    v1 += bf16_to_f32((unsigned short)(old[0] >> 16));
                    v2 += bf16_to_f32((unsigned short)(old[1] & 0xffffu));
                    v3 += bf16_to_f32((unsigned short)(old[1] >> 16));
                }
                u32x2 o;
                o[0] = pack_bf16x2(v0, v1);
                o[1] = pack_bf16x2(v2, v3);
                *dst = o;
            }
    }
    TRACE_STAMP(2);
}


// ---- v4: 8 waves, 256 queries per workgroup, one workgroup per CU, PING-PONG phases ------------------------------------
// v2 keeps two independent 4-wave workgroups per CU; the two waves that share a SIMD run the same loop in phase, so they
// fight for the matrix pipe during the S / PV products and leave it idle together during the softmax arithmetic
// (measured MFMA utilisation 35 %). Here the loop is software-pipelined across key tiles into two phases
//     X(t): O^T += V^T(t-1) . P^T(t-1)   then   S^T(t) = K(t) . Q^T        32 MFMAs, no exponentials
//     Y(t): row max, rescale decision, exp2, row sum, pack P^T(t)           VALU only
// separated by workgroup barriers, and waves 4..7 run ONE BARRIER BEHIND waves 0..3 (an extra barrier at their start, one
// at the others' end): of the two waves on a SIMD (w and w + 4) one is in X while the other is in Y. K and V^T tiles are
// shared by all 8 waves (half the LDS-DMA traffic of two 4-wave workgroups) in 3 + 3 slots of 16 KiB: at the start of
// X(t) every thread issues its share of K(t+2) and V^T(t+1); at the end of X(t) a counted vmcnt leaves only that group in
// flight, so a tile has 1.5-2 tile times to land and a slot is rewritten only after both halves finished reading it
// (K(t-1) and V^T(t-2) were last read by the lagging half in the interval before the leading half issues X(t)).
// The ragged last tile also comes by LDS-DMA: K rows >= Lk are fetched from row Lk-1 (their P is masked to 0), V^T
// chunks that would leave the row are fetched from its last chunk, and the thread that fetched a chunk zeroes its keys
// >= Lk in LDS right after its own vmcnt wait, before the barrier that publishes the tile (0 * garbage must be 0).
constexpr int NW4 = 8;
constexpr int QB4 = QW * NW4;             // 256 queries per workgroup
constexpr int SLOT = K_TILE_BYTES;        // 16 KiB; K slots 0..2, V^T slots 3..5
constexpr int V4_LDS = 6 * SLOT;          // 96 KiB

struct Dma4 {
    const char* kbase;      // K + h*D                         (uniform; tile t adds t*KT rows)
    const char* vbase;      // V^T + h*D rows                  (uniform; tile t adds t*KT columns)
    unsigned koff, voff;    // per-lane byte offsets of round 0
    int64_t krow;           // bytes per K row
    int64_t vrr;            // bytes between the two V^T rounds (64 rows)
    int kr;                 // K row of this lane in round 0 (round 1: + 32)
    int vc;                 // logical V^T chunk (8 keys) of this lane
};

__device__ __forceinline__ void dma4_init(Dma4& d, const AttnArgs& p, int h, int tid) {
    d.kr = tid >> 4;
    d.krow = p.ldk * 2;
    d.kbase = reinterpret_cast<const char*>(p.K + h * D);
    d.koff = (unsigned)(d.kr * d.krow) + (((tid & 15) ^ (d.kr & 15)) << 4);
    const int dd = tid >> 3;
    d.vc = (tid & 7) ^ ((dd >> 1) & 7);
    d.vbase = reinterpret_cast<const char*>(p.Vt + (int64_t)h * D * p.ldvt);
    d.voff = (unsigned)(dd * p.ldvt * 2) + (d.vc << 4);
    d.vrr = 64 * p.ldvt * 2;
}

__device__ __forceinline__ void dma4_k(const Dma4& d, const AttnArgs& p, int t, bool last_ragged, char* slot, int tid) {
    const char* base = d.kbase + (int64_t)t * KT * d.krow;
    char* l = slot + (tid >> 6) * 1024;
    if (!last_ragged) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(base + rr * 32 * d.krow + d.koff), (lds_void_t*)(l + rr * 8192), 16, 0, 0);
    } else {
        const int nrow = p.Lk - t * KT;       // 1..63 valid rows
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = d.kr + 32 * rr;
            const int rc = r < nrow ? r : nrow - 1;
            const char* g = base + (int64_t)rc * d.krow + (((tid & 15) ^ (d.kr & 15)) << 4);
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)g, (lds_void_t*)(l + rr * 8192), 16, 0, 0);
        }
    }
}

__device__ __forceinline__ void dma4_v(const Dma4& d, const AttnArgs& p, int t, bool last_ragged, char* slot, int tid) {
    const char* base = d.vbase + (int64_t)t * KT * 2;
    char* l = slot + (tid >> 6) * 1024;
    unsigned off = d.voff;
    if (last_ragged) {
        const int kc = t * KT + d.vc * 8;                         // first key of this lane's chunk
        const int kmax = (int)p.ldvt - 8;
        if (kc > kmax) off -= (unsigned)((kc - kmax) * 2);         // stay inside the row; such a chunk is zeroed afterwards
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(base + rr * d.vrr + off), (lds_void_t*)(l + rr * 8192), 16, 0, 0);
}

// keys >= Lk of the ragged last V^T tile -> 0, by the thread whose LDS-DMA brought the chunk (after its own vmcnt wait)
__device__ __forceinline__ void fix4_v(const Dma4& d, const AttnArgs& p, int t, char* slot, int tid) {
    const int nvalid = p.Lk - (t * KT + d.vc * 8);
    if (nvalid >= 8) return;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        u32x4* c = reinterpret_cast<u32x4*>(slot + rr * 8192 + tid * 16);
        u32x4 x = *c;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (2 * w >= nvalid) x[w] = 0u;
            else if (2 * w + 1 >= nvalid) x[w] &= 0xffffu;
        }
        *c = x;
    }
}

// exponentials + row sum + pack of one S tile against the exponent base m (S left untouched)
template <bool MASK>
__device__ __forceinline__ float exp_pack4(const f32x16 (&sacc)[2], bf16x8_t (&pf)[4], float m, const AttnArgs& p, int j0, int hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const f32x2 c2 = {p.scale_log2, p.scale_log2}, m2 = {-m, -m};
    f32x2 ps = {0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float pv[8];
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const int r = 8 * e + k;
                const f32x2 s2 = {sacc[b][r], sacc[b][r + 1]};
                const f32x2 x2 = s2 * c2 + m2;                      // v_pk_fma_f32
                f32x2 e2 = {__builtin_amdgcn_exp2f(x2[0]), __builtin_amdgcn_exp2f(x2[1])};
                if (MASK) {
                    const int key = j0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * hi;   // r even: r + 1 is the next key
                    e2[0] = key < p.Lk ? e2[0] : 0.f;
                    e2[1] = key + 1 < p.Lk ? e2[1] : 0.f;
                }
                ps += e2;                                            // v_pk_add_f32
                pv[k] = e2[0];
                pv[k + 1] = e2[1];
            }
            u32x4 w;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned ev = pack_bf16x2(pv[2 * i], pv[2 * i + 1]);           // group 2e
                const unsigned od = pack_bf16x2(pv[4 + 2 * i], pv[4 + 2 * i + 1]);   // group 2e+1
                const auto r = __builtin_amdgcn_permlane32_swap(ev, od, false, false);
                w[i] = r[0];
                w[2 + i] = r[1];
            }
            pf[2 * b + e] = __builtin_bit_cast(bf16x8_t, w);
        }
    return ps[0] + ps[1];
}

// Online-softmax step with an OPTIMISTIC exponent base: the exponentials are issued against the running base m_run of
// the previous tiles, so they do not wait for this tile's row-max reduction (a ~30-instruction dependent chain with a
// cross-half swap and a wave vote); the reduction runs beside them and only decides whether the tile has to be redone
// with a new base (some score exceeds the base by more than 2^8 — the first tile, then almost never). P <= 2^8 either way.
template <bool MASK>
__device__ __forceinline__ void softmax4(f32x16 (&sacc)[2], bf16x8_t (&pf)[4], f32x16 (&oacc)[4], float& m_run, float& l_run,
                                         const AttnArgs& p, int j0, int hi) {
    float psum = exp_pack4<MASK>(sacc, pf, m_run, p, j0, hi);
    float mx = sacc[0][0];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[b][r]);
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx * p.scale_log2);
    if (!__all(m_new - m_run <= DEFER_LOG2)) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        psum = exp_pack4<MASK>(sacc, pf, m_run, p, j0, hi);
    }
    l_run += psum;
}

// fragment rings of the X phase: VD V^T fragments / KD K fragment pairs in flight (the S accumulators are dead during
// the O^T products and the P fragments during S, so the registers are there; LDS latency under four reading waves is
// several MFMA times)
constexpr int VD = 4, KD = 3;   // deeper (8 / 4) measured 5 % slower
__device__ __forceinline__ void pv4(const char* vb, const bf16x8_t (&pf)[4], f32x16 (&oacc)[4], const int (&voff)[4]) {
    // product i: d-block db = i & 3, key group sg = i >> 2 — consecutive MFMAs write different accumulators
    __builtin_amdgcn_sched_barrier(0);
    bf16x8_t vf[VD];
#pragma unroll
    for (int i = 0; i < VD; ++i) vf[i] = *reinterpret_cast<const bf16x8_t*>(vb + voff[i >> 2] + (i & 3) * (32 * 128));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int db = i & 3, sg = i >> 2;
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i % VD], pf[sg], oacc[db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + VD < 16) {
            const int n = i + VD;
            vf[i % VD] = *reinterpret_cast<const bf16x8_t*>(vb + voff[n >> 2] + (n & 3) * (32 * 128));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__device__ __forceinline__ void s4(const char* kb, const bf16x8_t (&qf)[8], f32x16 (&sacc)[2], const int (&koff)[8]) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[b][r] = 0.f;
    bf16x8_t ka[KD], kc[KD];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KD - 1; ++ks) {
        ka[ks] = *reinterpret_cast<const bf16x8_t*>(kb + koff[ks]);
        kc[ks] = *reinterpret_cast<const bf16x8_t*>(kb + koff[ks] + 32 * 256);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if (ks + KD - 1 < 8) {
            ka[(ks + KD - 1) % KD] = *reinterpret_cast<const bf16x8_t*>(kb + koff[ks + KD - 1]);
            kc[(ks + KD - 1) % KD] = *reinterpret_cast<const bf16x8_t*>(kb + koff[ks + KD - 1] + 32 * 256);
            __builtin_amdgcn_sched_barrier(0);
        }
        sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks % KD], qf[ks], sacc[0], 0, 0, 0);
        sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[ks % KD], qf[ks], sacc[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

#define YUME_A4_BARRIER()                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0)

// steady-state tile: tile t (K slot KS = t % 3) with 1 <= t, t + 2 < number of full tiles — no ragged tile involved, both
// DMA groups present, every LDS slot address a compile-time constant
template <int KS>
__device__ __forceinline__ void steady4(const Dma4& dp, const AttnArgs& p, int t, char* smem, int tid, const bf16x8_t (&qf)[8],
                                        f32x16 (&sacc)[2], bf16x8_t (&pf)[4], f32x16 (&oacc)[4], float& m_run, float& l_run,
                                        int hi, const int (&koff)[8], const int (&voff)[4]) {
    dma4_k(dp, p, t + 2, false, smem + ((KS + 2) % 3) * SLOT, tid);
    dma4_v(dp, p, t + 1, false, smem + (3 + (KS + 1) % 3) * SLOT, tid);
    __builtin_amdgcn_s_setprio(1);
    pv4(smem + (3 + (KS + 2) % 3) * SLOT, pf, oacc, voff);
    s4(smem + KS * SLOT, qf, sacc, koff);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // everything older than this phase's 4 LDS-DMAs has landed
    YUME_A4_BARRIER();
    softmax4<false>(sacc, pf, oacc, m_run, l_run, p, t * KT, hi);
    YUME_A4_BARRIER();
}

// any tile: first, last, ragged, short groups (runtime slots)
__device__ __forceinline__ void general4(const Dma4& dp, const AttnArgs& p, int t, int nt, bool ragged, char* smem, int tid,
                                         const bf16x8_t (&qf)[8], f32x16 (&sacc)[2], bf16x8_t (&pf)[4], f32x16 (&oacc)[4],
                                         float& m_run, float& l_run, int hi, const int (&koff)[8], const int (&voff)[4]) {
    const int last = nt - 1;
    const bool has_k = t + 2 < nt, has_v = t + 1 < nt;
    if (has_k) dma4_k(dp, p, t + 2, ragged && t + 2 == last, smem + ((t + 2) % 3) * SLOT, tid);
    if (has_v) dma4_v(dp, p, t + 1, ragged && t + 1 == last, smem + (3 + (t + 1) % 3) * SLOT, tid);
    if (t > 0) pv4(smem + (3 + (t - 1) % 3) * SLOT, pf, oacc, voff);
    s4(smem + (t % 3) * SLOT, qf, sacc, koff);
    if (has_k) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (has_v) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ragged && t == last && nt > 1) fix4_v(dp, p, last, smem + (3 + last % 3) * SLOT, tid);
    YUME_A4_BARRIER();
    if (ragged && t == last)
        softmax4<true>(sacc, pf, oacc, m_run, l_run, p, t * KT, hi);
    else
        softmax4<false>(sacc, pf, oacc, m_run, l_run, p, t * KT, hi);
    YUME_A4_BARRIER();
}

__global__ __launch_bounds__(NW4 * 64, 2) void attn_fwd_kernel_v4(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[V4_LDS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                 // 0: leading half, 1: the half that runs one barrier behind
    const int hi = lane >> 5;
    const int ql = lane & 31;
    int h, qb;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int hx = (p.H + 7 - xcd) >> 3;
        const int per = hx * p.nqb;
        if (idx >= per) return;
        h = xcd + 8 * (idx / p.nqb);
        qb = idx % p.nqb;
    }
    const int q0 = p.q_lo + qb * QB4 + wave * QW;
    bf16x8_t qf[8];
    {
        int q = q0 + ql;
        q = q < p.Lq ? q : p.Lq - 1;
        const unsigned short* qp = p.Q + (int64_t)q * p.ldq + h * D + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * ks);
    }
    f32x16 oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;
    int koff[8], voff[4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) koff[ks] = ql * 256 + (((2 * ks + hi) ^ (ql & 15)) << 4);
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) voff[sg] = ql * 128 + (((2 * sg + hi) ^ ((ql >> 1) & 7)) << 4);
    Dma4 dp;
    dma4_init(dp, p, h, tid);

    const int nt = (p.Lk + KT - 1) / KT;
    const bool ragged = (p.Lk % KT) != 0;
    const int nfull = ragged ? nt - 1 : nt;
    const int last = nt - 1;
    // ---- prologue: K(0), K(1), V^T(0) ----
    dma4_k(dp, p, 0, ragged && last == 0, smem, tid);
    if (nt > 1) dma4_k(dp, p, 1, ragged && last == 1, smem + SLOT, tid);
    dma4_v(dp, p, 0, ragged && last == 0, smem + 3 * SLOT, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ragged && last == 0) fix4_v(dp, p, 0, smem + 3 * SLOT, tid);
    YUME_A4_BARRIER();
    if (grp == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

    f32x16 sacc[2];
    bf16x8_t pf[4];
    general4(dp, p, 0, nt, ragged, smem, tid, qf, sacc, pf, oacc, m_run, l_run, hi, koff, voff);
    int t = 1;
#pragma unroll 1
    for (; t + 4 < nfull; t += 3) {            // t % 3 == 1 here; the three calls issue tiles up to t + 4 (all full tiles)
        steady4<1>(dp, p, t, smem, tid, qf, sacc, pf, oacc, m_run, l_run, hi, koff, voff);
        steady4<2>(dp, p, t + 1, smem, tid, qf, sacc, pf, oacc, m_run, l_run, hi, koff, voff);
        steady4<0>(dp, p, t + 2, smem, tid, qf, sacc, pf, oacc, m_run, l_run, hi, koff, voff);
    }
#pragma unroll 1
    for (; t < nt; ++t) general4(dp, p, t, nt, ragged, smem, tid, qf, sacc, pf, oacc, m_run, l_run, hi, koff, voff);
    // ================= X(nt): the last tile's O^T products =================
    pv4(smem + (3 + last % 3) * SLOT, pf, oacc, voff);
    if (grp == 0) __builtin_amdgcn_s_barrier();

    const float l_tot = xhalf_sum(l_run);
    const float inv = 1.0f / l_tot;
    const int q = q0 + ql;
    if (q < p.Lq) {
        unsigned short* op = p.O + (int64_t)q * p.ldo + h * D + 4 * hi;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = oacc[db][4 * g + 0] * inv, v1 = oacc[db][4 * g + 1] * inv;
                float v2 = oacc[db][4 * g + 2] * inv, v3 = oacc[db][4 * g + 3] * inv;
                u32x2* dst = reinterpret_cast<u32x2*>(op + 32 * db + 8 * g);
                if (p.accumulate) {
                    const u32x2 old = *dst;
                    v0 += bf16_to_f32((unsigned short)(old[0] & 0xffffu));
                    v1 += bf16_to_f32((unsigned short)(old[0] >> 16));
                    v2 += bf16_to_f32((unsigned short)(old[1] & 0xffffu));
                    v3 += bf16_to_f32((unsigned short)(old[1] >> 16));
                }
                u32x2 o;
                o[0] = pack_bf16x2(v0, v1);
                o[1] = pack_bf16x2(v2, v3);
                *dst = o;
            }
    }
}


// merge the key-range splits of the v2 kernel: O = sum_s O_s 2^(m_s - m) / sum_s l_s 2^(m_s - m), m = max_s m_s (fixed order)
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, int splits,
                                                           int64_t rows, int H, unsigned short* __restrict__ O, int64_t ldo, int q_lo,
                                                           int accumulate) {
    const int64_t nq = rows * H * (D / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (D / 4));
        const int h = (int)((i / (D / 4)) % H);
        const int64_t r = i / ((int64_t)(D / 4) * H);
        float m = NEG_BIG;
        for (int s = 0; s < splits; ++s) m = fmaxf(m, part_ml[((s * rows + r) * H + h) * 2]);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float l = 0.f;
        for (int s = 0; s < splits; ++s) {
            const float* ml = part_ml + ((s * rows + r) * H + h) * 2;
            const float a = __builtin_amdgcn_exp2f(ml[0] - m);
            l += ml[1] * a;
            acc += *reinterpret_cast<const f32x4*>(part_o + (s * rows + r) * ((int64_t)H * D) + h * D + 4 * c4) * a;
        }
        const float inv = 1.0f / l;
        float v0 = acc[0] * inv, v1 = acc[1] * inv, v2 = acc[2] * inv, v3 = acc[3] * inv;
        u32x2* dst = reinterpret_cast<u32x2*>(O + (q_lo + r) * ldo + h * D + 4 * c4);
        if (accumulate) {
            const u32x2 old = *dst;
            v0 += bf16_to_f32((unsigned short)(old[0] & 0xffffu));
            v1 += bf16_to_f32((unsigned short)(old[0] >> 16));
            v2 += bf16_to_f32((unsigned short)(old[1] & 0xffffu));
            v3 += bf16_to_f32((unsigned short)(old[1] >> 16));
        }
        u32x2 o;
        o[0] = pack_bf16x2(v0, v1);
        o[1] = pack_bf16x2(v2, v3);
        *dst = o;
    }
}

}  // namespace


// ---- launch plan of the one-wave-per-SIMD kernel (attn_fwd7.hip) ------------------------------------------------------------
// An XCD owns ceil(H/8) heads x nq query blocks of 256 rows = that many workgroups of equal length on its 32 CUs; a partial
// last round costs a whole round. The last `tail_q` query blocks of every head can instead be cut into `splits` key ranges
// whose pieces are dispatched behind the whole blocks (same launch) and merged by attn_combine_kernel. The plan minimises the
// makespan of that in-order dispatch under a simple cost model (a piece = 1/splits of a block + a fixed prologue share).
struct Plan7 { int64_t tail_qb; int splits; };
static Plan7 attn7_plan_search(int64_t Lq, int64_t Lk, int64_t H);
// the search below (~37 makespan simulations with a 32-way min-scan per workgroup) is a pure function of the launch shape and
// runs on the host inside every yume_attn_fwd_ws call (30-40 times per denoise step): memoised per calling thread
static Plan7 attn7_plan(int64_t Lq, int64_t Lk, int64_t H) {
    struct Entry { int64_t Lq, Lk, H; Plan7 pl; };
    static thread_local Entry memo[8];
    static thread_local int used = 0, next = 0;
    for (int i = 0; i < used; ++i)
        if (memo[i].Lq == Lq && memo[i].Lk == Lk && memo[i].H == H) return memo[i].pl;
    const Plan7 pl = attn7_plan_search(Lq, Lk, H);
    memo[next] = Entry{Lq, Lk, H, pl};
    next = (next + 1) & 7;
    used = used < 8 ? used + 1 : 8;
    return pl;
}
static Plan7 attn7_plan_search(int64_t Lq, int64_t Lk, int64_t H) {
    const int64_t nq = (Lq + QB4 - 1) / QB4, hx = (H + 7) / 8, nt = (Lk + KT - 1) / KT;
    Plan7 best{nq, 1};
    auto makespan = [&](int64_t tail_q, int splits) {
        double cu[32];
        for (double& c : cu) c = 0.0;
        auto put = [&](double cost) {
            int m = 0;
            for (int i = 1; i < 32; ++i) if (cu[i] < cu[m]) m = i;
            cu[m] += cost;
        };
        for (int64_t i = 0; i < hx * (nq - tail_q); ++i) put(1.0);
        for (int64_t i = 0; i < hx * tail_q * splits; ++i) put(1.0 / splits + 0.04);
        double mx = 0.0;
        for (double c : cu) mx = c > mx ? c : mx;
        return mx + (splits > 1 ? 0.03 : 0.0);       // + the merge pass
    };
    double bm = makespan(0, 1);
    for (int splits = 2; splits <= 4; ++splits) {
        if (nt / splits < 16) break;
        for (int64_t tail_q = 1; tail_q <= nq && tail_q <= 12; ++tail_q) {
            const double m = makespan(tail_q, splits);
            if (m < bm - 0.02) { bm = m; best = Plan7{nq - tail_q, splits}; }
        }
    }
    return best;
}
static bool attn7_applies(int64_t Lq, int64_t Lk) { return Lk >= 1536 && Lq >= QB4; }

// ---- launch plan of the persistent kernel (attn_fwd8.hip) ------------------------------------------------------------------
// The same item list as attn_fwd7's (whole query blocks, then the key-range pieces of the last `nq - tail_qb` blocks of every head), drawn
// by ticket instead of dispatched in block-id order. An item boundary costs ~2.5 tile times there (two bubbles) instead of a whole prologue
// and epilogue, so shorter pieces pay: down to 8 key tiles (what the GPU tests exercise; the kernel's own protocol needs 5).
static Plan7 attn8_plan_search(int64_t Lq, int64_t Lk, int64_t H) {
    const int64_t nq = (Lq + QB4 - 1) / QB4, hx = (H + 7) / 8, nt = (Lk + KT - 1) / KT;
    Plan7 best{nq, 1};
    const double bub = 2.5, whole = (double)nt + bub;
    auto makespan = [&](int64_t tail_q, int splits) {
        double cu[32];
        for (double& c : cu) c = 0.0;
        auto put = [&](double cost) {
            int m = 0;
            for (int i = 1; i < 32; ++i) if (cu[i] < cu[m]) m = i;
            cu[m] += cost;
        };
        for (int64_t i = 0; i < hx * (nq - tail_q); ++i) put(whole);
        for (int64_t i = 0; i < hx * tail_q * splits; ++i) put((double)nt / splits + bub + 1.0);        // + the fp32 partial store
        double mx = 0.0;
        for (double c : cu) mx = c > mx ? c : mx;
        return mx + (splits > 1 ? 6.0 : 0.0);       // + the merge pass
    };
    double bm = makespan(0, 1);
    for (int splits = 2; splits <= 4; ++splits) {
        if (nt / splits < 8) break;                  // (an item of the stream is at least 8 tiles: its successor's ticket is drawn while it runs, with room)
        for (int64_t tail_q = 1; tail_q <= nq && tail_q <= 12; ++tail_q) {
            const double m = makespan(tail_q, splits);
            if (m < bm - 0.01 * whole) { bm = m; best = Plan7{nq - tail_q, splits}; }
        }
    }
    return best;
}
static Plan7 attn8_plan(int64_t Lq, int64_t Lk, int64_t H) {
    struct Entry { int64_t Lq, Lk, H; Plan7 pl; };
    static thread_local Entry memo[8];
    static thread_local int used = 0, next = 0;
    for (int i = 0; i < used; ++i)
        if (memo[i].Lq == Lq && memo[i].Lk == Lk && memo[i].H == H) return memo[i].pl;
    const Plan7 pl = attn8_plan_search(Lq, Lk, H);
    memo[next] = Entry{Lq, Lk, H, pl};
    next = (next + 1) & 7;
    used = used < 8 ? used + 1 : 8;
    return pl;
}
// shapes the persistent kernel takes (the caller's flags and the counter workspace are checked at the call)
static bool attn8_applies(int64_t Lq, int64_t Lk) { return Lk >= 8 * KT && Lq >= QB4; }
static bool attn8_enabled() {
    static const bool on = [] { const char* v = getenv("YUME_ATTN_V8"); return !v || atoi(v) != 0; }();
    return on;
}
static int cu_count() {
    static thread_local int n[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!n[dev]) {
        int v = 0;
        n[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return n[dev];
}

static int64_t attn7_workspace(int64_t Lq, int64_t Lk, int64_t H);
static int64_t attn8_workspace(int64_t Lq, int64_t Lk, int64_t H);
// (the call does not know which flags the launches will carry: the larger of the two kernels' needs)
extern "C" int64_t yume_attn_workspace_bytes(int64_t Lq, int64_t Lk, int64_t H) {
    if (Lq <= 0 || Lk <= 0 || H <= 0) return 0;
    const int64_t a = attn7_workspace(Lq, Lk, H), b = attn8_workspace(Lq, Lk, H);
    return a > b ? a : b;
}
static int64_t attn7_workspace(int64_t Lq, int64_t Lk, int64_t H) {
    if (!attn7_applies(Lq, Lk)) return 0;
    const Plan7 pl = attn7_plan(Lq, Lk, H);
    if (pl.splits == 1) return 0;
    const int64_t rows = Lq - pl.tail_qb * QB4;
    return (int64_t)pl.splits * rows * (H * D + H * 2) * 4;
}
static int64_t attn8_workspace(int64_t Lq, int64_t Lk, int64_t H) {
    if (!attn8_applies(Lq, Lk)) return 0;
    const Plan7 pl = attn8_plan(Lq, Lk, H);
    if (pl.splits == 1) return 0;
    return (int64_t)pl.splits * (Lq - pl.tail_qb * QB4) * (H * D + H * 2) * 4;
}

extern "C" int yume_attn_fwd_ws(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* Vt, int64_t ldvt,
                                void* O, int64_t ldo, int64_t Lq, int64_t Lk, int64_t H, float scale, int accumulate,
                                int variant, void* workspace, int64_t workspace_bytes, void* stream);

extern "C" int yume_attn_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* Vt, int64_t ldvt,
                             void* O, int64_t ldo, int64_t Lq, int64_t Lk, int64_t H, float scale, int accumulate,
                             int variant, void* stream) {
    return yume_attn_fwd_ws(Q, ldq, K, ldk, Vt, ldvt, O, ldo, Lq, Lk, H, scale, accumulate, variant, nullptr, 0, stream);
}

extern "C" int yume_attn_fwd_ws(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* Vt, int64_t ldvt,
                                void* O, int64_t ldo, int64_t Lq, int64_t Lk, int64_t H, float scale, int accumulate,
                                int variant, void* workspace, int64_t workspace_bytes, void* stream) {
    YUME_REQUIRE(Q && K && Vt && O, "attn_fwd: NULL pointer");
    YUME_REQUIRE(Lq > 0 && Lk > 0 && H > 0, "attn_fwd: empty problem Lq=%lld Lk=%lld H=%lld", (long long)Lq, (long long)Lk, (long long)H);
    YUME_REQUIRE(Lq < (1ll << 30) && Lk < (1ll << 30) && H < 65536, "attn_fwd: dimension too large");
    YUME_REQUIRE((ldq % 8) == 0 && (ldk % 8) == 0 && (ldvt % 8) == 0 && (ldo % 4) == 0, "attn_fwd: ldq/ldk/ldvt must be multiples of 8, ldo of 4");
    YUME_REQUIRE(ldk < (1ll << 24) && ldvt < (1ll << 24), "attn_fwd: ldk / ldvt too large for 32-bit tile offsets");
    YUME_REQUIRE(ldvt >= Lk && ldvt >= 8, "attn_fwd: ldvt=%lld must be >= Lk=%lld", (long long)ldvt, (long long)Lk);
    YUME_REQUIRE(((uintptr_t)Q % 16) == 0 && ((uintptr_t)K % 16) == 0 && ((uintptr_t)Vt % 16) == 0 && ((uintptr_t)O % 8) == 0, "attn_fwd: pointer alignment");
    AttnArgs a;
    a.Q = (const unsigned short*)Q; a.ldq = ldq;
    a.K = (const unsigned short*)K; a.ldk = ldk;
    a.Vt = (const unsigned short*)Vt; a.ldvt = ldvt;
    a.O = (unsigned short*)O; a.ldo = ldo;
    a.Lq = (int)Lq; a.Lk = (int)Lk; a.H = (int)H;
    // YUME_ATTN_Q_PRESCALED: Q already carries scale * log2(e) (the caller folded it into the producer of Q before its bf16 rounding): the
    // scores are the exponents. `scale` is ignored; every kernel sees scale_log2 = 1, the one-wave-per-SIMD kernel runs its base-free pieces.
    const int q_pre = (variant & YUME_ATTN_Q_PRESCALED) != 0, kv_pad = (variant & YUME_ATTN_KV_PADDED) != 0;
    variant &= ~(YUME_ATTN_Q_PRESCALED | YUME_ATTN_KV_PADDED);
    a.q_prescaled = q_pre;
    a.scale_log2 = q_pre ? 1.0f : scale * 1.4426950408889634f;
    a.accumulate = accumulate;
    a.q_lo = 0;
    a.nqb = 0;
    a.part_o = nullptr;
    a.part_ml = nullptr;
    a.tail_qb = 0;
    a.splits = 1;
    hipStream_t st = (hipStream_t)stream;
    // launch one kernel over the query rows [lo, hi)
    auto run = [&](int kernel, int64_t lo, int64_t hi) {
        AttnArgs b = a;
        b.q_lo = (int)lo;
        b.Lq = (int)hi;
        const int qb = (kernel == 4 || kernel == 7) ? QB4 : QB;
        b.nqb = (int)((hi - lo + qb - 1) / qb);
        b.tail_qb = b.nqb;
        b.splits = 1;
        // every XCD slot gets ceil(H/8)*nqb block ids; surplus ids exit immediately
        const dim3 grid((unsigned)(((H + 7) / 8) * b.nqb * 8));
        if (kernel == 7)
            yume_attn7_launch(b, st);
        else if (kernel == 4)
            hipLaunchKernelGGL(attn_fwd_kernel_v4, grid, dim3(NW4 * 64), 0, st, b);
        else if (kernel == 1)
            hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(NW * 64), 0, st, b);
        else
            hipLaunchKernelGGL(attn_fwd_kernel_v2, grid, dim3(NW * 64), 0, st, b);
    };
    // the persistent kernel (attn_fwd8.hip): base-free body only, K / V^T padded to whole key tiles (the caller's word: YUME_ATTN_KV_PADDED;
    // ldvt can be checked), a registered counter workspace for its tickets. variant 8 insists on it, variant 0 takes it where it applies.
    const int64_t nt8 = (Lk + KT - 1) / KT;
    const bool v8_fits = q_pre && kv_pad && attn8_applies(Lq, Lk) && ldvt >= nt8 * KT &&
                         Lq * ldq * 2 + 512 < (1ll << 32);      // (its Q' loads address a query row by a 32-bit byte offset from the head's base)
    if (variant == 8) {
        YUME_REQUIRE(v8_fits, "attn_fwd: variant 8 needs YUME_ATTN_Q_PRESCALED | YUME_ATTN_KV_PADDED, Lk >= 512, Lq >= 256 and ldvt >= %lld",
                     (long long)(nt8 * KT));
    }
    // (variant 0: where attn_fwd7 was the choice. Measured, the 512-key cross-attention — 8 tiles per item, an item boundary every 12 us —
    // stays faster on the 4-wave kernel: 0.101 against 0.132 ms in the bench, profiles/r4_bench_ab_v8_on_off.log)
    if (variant == 8 || (variant == 0 && v8_fits && attn7_applies(Lq, Lk) && attn8_enabled())) {
        int* counters = yume_counters::next_set();
        if (variant == 8) YUME_REQUIRE(counters != nullptr, "attn_fwd: variant 8 needs a registered counter workspace (yume_counter_workspace_init)");
        if (counters) {
            Plan7 pl = attn8_plan(Lq, Lk, H);
            const int64_t rows = Lq - pl.tail_qb * QB4;
            const int64_t need = pl.splits > 1 ? (int64_t)pl.splits * rows * (H * D + H * 2) * 4 : 0;
            AttnArgs b = a;
            b.nqb = (int)((Lq + QB4 - 1) / QB4);
            if (pl.splits > 1 && workspace && workspace_bytes >= need) {
                b.tail_qb = (int)pl.tail_qb;
                b.splits = pl.splits;
                b.part_o = reinterpret_cast<float*>(workspace);
                b.part_ml = b.part_o + (int64_t)pl.splits * rows * H * D;
            } else {
                pl = Plan7{b.nqb, 1};              // no scratch: whole query blocks only
                b.tail_qb = b.nqb;
                b.splits = 1;
            }
            int64_t items = 0;
            for (int y = 0; y < 8; ++y) items += ((H + 7 - y) >> 3) * (b.tail_qb + (int64_t)(b.nqb - b.tail_qb) * b.splits);
            const int nwg = (int)(items < cu_count() ? items : cu_count());
            yume_attn8_launch(b, counters, nwg, st);
            if (b.splits > 1) {
                const int64_t nq = rows * H * (D / 4);
                hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, b.part_o, b.part_ml, b.splits, rows,
                                   (int)H, (unsigned short*)O, ldo, (int)(b.tail_qb * QB4), accumulate);
            }
            YUME_CHECK_LAUNCH("attn_fwd");
            return YUME_OK;
        }
    }
    if (variant == 9) YUME_REQUIRE(attn_rk::fits(Lq, Lk, ldvt, kv_pad), "attn_fwd: variant 9 needs 448 < Lk <= 512 (padded), Lq >= 1024, ldvt >= 512");
    if (variant == 9 || (variant == 0 && attn_rk::applies(Lq, Lk, ldvt, kv_pad))) {
        // the 512-key cross-attention: K and V^T resident in registers, persistent workgroups (attn_cross_rk.hpp, r6)
        attn_rk::launch(a, cu_count(), st);
        YUME_CHECK_LAUNCH("attn_fwd");
        return YUME_OK;
    }
    if (variant == 1 || variant == 2 || variant == 4 || variant == 7) {
        run(variant, 0, Lq);
    } else if (!attn7_applies(Lq, Lk)) {
        run(2, 0, Lq);           // few key tiles / few queries: the 4-wave kernel's shorter prologue and smaller blocks win (cross-attention)
    } else {
        // one-wave-per-SIMD kernel, one 256-query workgroup per CU; the blocks of a partial last round are cut into key ranges
        // when the caller gave the scratch for their partial results
        const Plan7 pl = attn7_plan(Lq, Lk, H);
        const int64_t rows = Lq - pl.tail_qb * QB4;
        const int64_t need = pl.splits > 1 ? (int64_t)pl.splits * rows * (H * D + H * 2) * 4 : 0;
        if (pl.splits > 1 && workspace && workspace_bytes >= need) {
            AttnArgs b = a;
            b.nqb = (int)((Lq + QB4 - 1) / QB4);
            b.tail_qb = (int)pl.tail_qb;
            b.splits = pl.splits;
            b.part_o = reinterpret_cast<float*>(workspace);
            b.part_ml = b.part_o + (int64_t)pl.splits * rows * H * D;
            yume_attn7_launch(b, st);
            const int64_t nq = rows * H * (D / 4);
            hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, b.part_o, b.part_ml, pl.splits,
                               rows, (int)H, (unsigned short*)O, ldo, (int)(pl.tail_qb * QB4), accumulate);
        } else {
            run(7, 0, Lq);
        }
    }
    YUME_CHECK_LAUNCH("attn_fwd");
    return YUME_OK;
}
