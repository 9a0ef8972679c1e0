// attn_fwd.hip — exact-softmax attention forward for head_dim 128, bf16 in / fp32 accumulate / bf16 out (gfx950).
//
// One workgroup = 4 waves = 128 queries of one head; each wave owns 32 queries and walks the
// keys in tiles of 64. Everything is computed TRANSPOSED so that the softmax row of a query
// lives in ONE lane (plus its partner lane^32) and never needs LDS or cross-lane shuffles:
//
//   S^T[key, q] = K[key, :] . Q[q, :]        v_mfma_f32_32x32x16_bf16, A = K tile (LDS), B = Q^T (registers)
//                 C layout: col = lane&31 = q, row = key = (r&3) + 8*(r>>2) + 4*(lane>>5)
//   P^T         = exp2(S^T*c - m)            in registers; packed to bf16 it IS the B operand of
//   O^T[d, q]   = V^T[d, key] . P^T[key, q]  A = V^T tile (LDS, K-major image), B = P^T (registers)
//                 with the SAME key<->k-slot assignment on both operands, so no permutation is needed.
//   O^T accumulators keep col = lane&31 = q, so the online-softmax rescale is lane-local too.
//
// K tile  : LDS [64 keys][128 d] bf16, 16-byte chunk c of row r stored at chunk c ^ (r & 15)  (ds_read_b128 conflict-free)
// V^T tile: LDS [128 d][64 keys] bf16, row stride 136 B                                     (ds_read_b64 conflict-free)
// Both tiles are register-staged (global -> VGPR issued one tile ahead, VGPR -> LDS after the
// compute of the current tile) and double buffered in LDS: one barrier per key tile.
// Roofline: MFMA (bf16 dense). Algorithmic work 4*Lq*Lk*128 flop per head.
#include "common.hpp"

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

struct AttnArgs {
    const unsigned short* Q; int64_t ldq;
    const unsigned short* K; int64_t ldk;
    const unsigned short* Vt; int64_t ldvt;
    unsigned short* O; int64_t ldo;
    int Lq, Lk, H;
    float scale_log2;  // softmax scale * log2(e)
    int accumulate;
    int nqb;           // query blocks per head
};

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
    const int q0 = qb * QB + wave * QW;

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
    const int q0 = qb * QB + wave * QW;
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
    if (nt == 1 && ragged) {
        Stage st;
        stage_load(st, p, h, 0, tid);
        stage_store_v2(st, smem, tid);
    } else {
        dma_tile(dp, kstep, smem, wave);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        char* kb = smem + cur * V2_BUF;
        char* nb = smem + (cur ^ 1) * V2_BUF;
        const bool has_next = t + 1 < nt;
        const bool next_reg = has_next && ragged && (t + 2 == nt);
        if (has_next && !next_reg) dma_tile(dp, kstep, nb, wave);
        if (!has_next && ragged)
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

}  // namespace


extern "C" int yume_attn_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* Vt, int64_t ldvt,
                             void* O, int64_t ldo, int64_t Lq, int64_t Lk, int64_t H, float scale, int accumulate,
                             int variant, void* stream) {
    YUME_REQUIRE(Q && K && Vt && O, "attn_fwd: NULL pointer");
    YUME_REQUIRE(Lq > 0 && Lk > 0 && H > 0, "attn_fwd: empty problem Lq=%lld Lk=%lld H=%lld", (long long)Lq, (long long)Lk, (long long)H);
    YUME_REQUIRE(Lq < (1ll << 30) && Lk < (1ll << 30) && H < 65536, "attn_fwd: dimension too large");
    YUME_REQUIRE((ldq % 8) == 0 && (ldk % 8) == 0 && (ldvt % 8) == 0 && (ldo % 4) == 0, "attn_fwd: ldq/ldk/ldvt must be multiples of 8, ldo of 4");
    YUME_REQUIRE(ldvt >= Lk && ldvt >= 8, "attn_fwd: ldvt=%lld must be >= Lk=%lld", (long long)ldvt, (long long)Lk);
    YUME_REQUIRE(((uintptr_t)Q % 16) == 0 && ((uintptr_t)K % 16) == 0 && ((uintptr_t)Vt % 16) == 0 && ((uintptr_t)O % 8) == 0, "attn_fwd: pointer alignment");
    AttnArgs a;
    a.Q = (const unsigned short*)Q; a.ldq = ldq;
    a.K = (const unsigned short*)K; a.ldk = ldk;
    a.Vt = (const unsigned short*)Vt; a.ldvt = ldvt;
    a.O = (unsigned short*)O; a.ldo = ldo;
    a.Lq = (int)Lq; a.Lk = (int)Lk; a.H = (int)H;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.accumulate = accumulate;
    a.nqb = (int)((Lq + QB - 1) / QB);
    // every XCD slot gets ceil(H/8)*nqb block ids; surplus ids exit immediately
    const int64_t per_xcd = ((H + 7) / 8) * a.nqb;
    dim3 grid((unsigned)(per_xcd * 8)), block(NW * 64);
    if (variant == 1)
        hipLaunchKernelGGL(attn_fwd_kernel, grid, block, 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(attn_fwd_kernel_v2, grid, block, 0, (hipStream_t)stream, a);
    YUME_CHECK_LAUNCH("attn_fwd");
    return YUME_OK;
}
