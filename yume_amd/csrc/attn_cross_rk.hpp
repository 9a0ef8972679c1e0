// attn_cross_rk.hpp — cross-attention over a SHORT key sequence (256 < Lk <= 512: the 512 text tokens of wan23/modules/model.py:210-232,
// the 257 CLIP tokens of wan/modules/model.py:348-389) with the head's K and V^T RESIDENT IN REGISTERS for a workgroup's whole share of the
// launch (r6).
//
// Both earlier designs stream K / V^T tiles through LDS once per query block: on the 4-wave kernel (attn_fwd_kernel_v2) a 128-query block
// re-stages all 8 tiles (prologue + epilogue = 20 % of a launch, 3.47 rounds of workgroups quantised to 4, fabric traffic 2.3 x algorithmic),
// on the persistent kernel an item boundary comes every 12 us (profiles/r5_trace_attention_v8_cross_shape.md). With 512 keys the operands are
// small enough to stay put: one workgroup = 4 waves, one per SIMD, 512 registers per lane;
//   * S phase: wave w owns keys [128 w, 128 w + 128): its K fragments (4 key blocks x 8 k-steps of v_mfma_f32_32x32x16_bf16 A operands =
//     128 registers) are loaded once per head; S^T = K Q^T for a block of 32 queries is 32 MFMAs;
//   * softmax: exact, two barriers per query block — the four waves exchange their row maxima through LDS (512 B), then each wave writes
//     its exponentials as bf16 P^T fragments (32 KiB per block) and its partial row sums;
//   * O phase: wave w owns output features d in [32 w, 32 w + 32) over ALL 512 keys: its V^T fragments (32 key steps = 128 registers) are
//     loaded once per head, the P^T B operands are read back from LDS fragment by fragment: O^T[32 d x 32 q] += V^T P^T is 32 MFMAs and is
//     COMPLETE in the wave — no partial (m, l, O) merge;
//   * P^T is exchanged in MFMA fragment order: lane (q, half) of the producer writes the 16 bytes lane (q, half) of every consumer reads
//     (the K rows are loaded in a permuted order — MFMA row m takes key m with bits 2 and 3 exchanged — so that the accumulator holds eight
//     consecutive keys per lane half and 16-key step: the V^T fragments are plain contiguous 16-byte loads; no v_permlane32_swap,
//     conflict-free 1 KiB reads);
//   * workgroups are persistent over a contiguous range of (head, query block) units; Q streams straight from global memory into B
//     fragments, one block ahead.
// Bound: MFMA (4 Lq Lk 128 flop per head). The compiler schedules the code; nothing here names registers.
#pragma once
#include "common.hpp"
#include "attn_args.hpp"

namespace attn_rk {

constexpr int HD = 128;            // head dim
constexpr int QB = 32;             // queries per unit
constexpr int LKMAX = 512;
constexpr int P_BYTES = 32 * 64 * 16;   // P^T of one unit: 32 key steps x 64 lanes x 16 B
constexpr int LDS_BYTES = 2 * P_BYTES + 2 * 2 * 4 * 32 * 4;

__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <bool MASK>
__global__ __launch_bounds__(256, 1) void attn_cross_rk_kernel(AttnArgs p, int nqb, int nunit) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    float* const mbuf = reinterpret_cast<float*>(smem + 2 * P_BYTES);          // [2][4][32]
    float* const lbuf = mbuf + 2 * 4 * 32;                                       // [2][4][32]

    // this workgroup's units [u0, u1) of the head-major (head, query block) order
    const int u0 = (int)(((int64_t)blockIdx.x * nunit) / gridDim.x), u1 = (int)(((int64_t)(blockIdx.x + 1) * nunit) / gridDim.x);
    if (u0 >= u1) return;

    bf16x8_t kf[4][8], vf[32];
    int head = -1;
    auto load_q = [&](int u, bf16x8_t (&qf)[8]) {
        const int h = u / nqb, qb = u - h * nqb;
        int q = p.q_lo + qb * QB + ql;
        q = q < p.Lq ? q : p.Lq - 1;
        const unsigned short* qp = p.Q + (int64_t)q * p.ldq + h * HD + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * ks);
    };
    bf16x8_t qf[8];
    load_q(u0, qf);

    for (int u = u0; u < u1; ++u) {
        const int h = u / nqb, qb = u - h * nqb;
        const int par = u & 1;
        if (h != head) {                                                          // (uniform) the head's K and V^T fragments of this wave
            head = h;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                // MFMA row ql of the block takes key kperm(ql) = ql with bits 2 and 3 exchanged: the accumulator then holds, per lane half and
                // 16-key step, EIGHT CONSECUTIVE keys (16 st + 8 hh + 0..7) — the P^T fragment is the accumulator's own order and the V^T
                // fragment one contiguous 16-byte load (first build: keys 4 hh + (0..3) and 8 + 4 hh + (0..3), two 8-byte loads a row apart per lane)
                int key = 128 * wave + 32 * kb + ((ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1));
                key = key < p.Lk ? key : p.Lk - 1;                                // (rows beyond Lk: masked below)
                const unsigned short* kp = p.K + (int64_t)key * p.ldk + h * HD + 8 * hh;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) kf[kb][ks] = *reinterpret_cast<const bf16x8_t*>(kp + 16 * ks);
            }
            // V^T rows d = 32 wave + ql: the 8 keys 16 st + 8 hh + (0..7) of step st (the caller pads V^T with finite values up to a whole
            // number of 64-key tiles: YUME_ATTN_KV_PADDED; columns beyond Lk meet p = 0)
            const unsigned short* vp = p.Vt + (int64_t)(h * HD + 32 * wave + ql) * p.ldvt + 8 * hh;
#pragma unroll
            for (int st = 0; st < 32; ++st) vf[st] = *reinterpret_cast<const bf16x8_t*>(vp + 16 * st);
        }
        // ---- S^T = K Q^T: 4 key blocks x 8 k-steps (the four accumulators alternate: a dependent MFMA comes back four MFMAs later)
        f32x16 s[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][ks], qf[ks], s[kb], 0, 0, 0);
        if (u + 1 < u1) load_q(u + 1, qf);                                       // next unit's Q, one block ahead
        // ---- row maximum over this wave's 128 keys, then over the four waves
        float mx = -3.0e38f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = s[kb][r] * p.scale_log2;
                if (MASK) {
                    const int key = 128 * wave + 32 * kb + 16 * (r >> 3) + 8 * hh + (r & 7);
                    x = key < p.Lk ? x : -3.0e38f;
                }
                s[kb][r] = x;
                mx = fmaxf(mx, x);
            }
        mx = xhalf_max(mx);
        if (hh == 0) mbuf[(par * 4 + wave) * 32 + ql] = mx;
        __syncthreads();                                                          // barrier A
        const float m = fmaxf(fmaxf(mbuf[(par * 4 + 0) * 32 + ql], mbuf[(par * 4 + 1) * 32 + ql]),
                              fmaxf(mbuf[(par * 4 + 2) * 32 + ql], mbuf[(par * 4 + 3) * 32 + ql]));
        // ---- exponentials, partial row sum, P^T fragments (the accumulator's own key order) -> LDS
        float lsum = 0.f;
        char* const pw = smem + par * P_BYTES + (wave * 8) * 1024 + lane * 16;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    e[j] = __builtin_amdgcn_exp2f(s[kb][8 * st + j] - m);
                    lsum += e[j];
                }
                u32x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = pack_bf16x2(e[2 * j], e[2 * j + 1]);
                *reinterpret_cast<u32x4*>(pw + (kb * 2 + st) * 1024) = w;
            }
        lsum = xhalf_sum(lsum);
        if (hh == 0) lbuf[(par * 4 + wave) * 32 + ql] = lsum;
        __syncthreads();                                                          // barrier B
        // ---- O^T[32 d of this wave x 32 q] = V^T P^T over all 512 keys. Four partial accumulators (one chain of 32 dependent MFMAs would run
        // at half rate) and an 8-deep ring of P^T fragments, pinned with sched_barriers: left alone, the scheduler emits
        // `ds_read ; s_waitcnt lgkmcnt(0) ; v_mfma` thirty-two times through one register quad (first build: 4.1 us per unit)
        f32x16 oa[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) oa[a][r] = 0.f;
        const char* const pr = smem + par * P_BYTES + lane * 16;
        bf16x8_t pf[8];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = *reinterpret_cast<const bf16x8_t*>(pr + j * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < 32; ++st) {
            oa[st & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st], pf[st & 7], oa[st & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (st + 8 < 32) {
                pf[st & 7] = *reinterpret_cast<const bf16x8_t*>(pr + (st + 8) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = (oa[0][r] + oa[1][r]) + (oa[2][r] + oa[3][r]);
        const float l = (lbuf[(par * 4 + 0) * 32 + ql] + lbuf[(par * 4 + 1) * 32 + ql]) + (lbuf[(par * 4 + 2) * 32 + ql] + lbuf[(par * 4 + 3) * 32 + ql]);
        const float inv = 1.0f / l;
        const int q = p.q_lo + qb * QB + ql;
        if (q < p.Lq) {
            unsigned short* op = p.O + (int64_t)q * p.ldo + h * HD + 32 * wave + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = o[4 * g + 0] * inv, v1 = o[4 * g + 1] * inv, v2 = o[4 * g + 2] * inv, v3 = o[4 * g + 3] * inv;
                u32x2* dst = reinterpret_cast<u32x2*>(op + 8 * g);
                if (p.accumulate) {
                    const u32x2 old = *dst;
                    v0 += bf16_to_f32((unsigned short)(old[0] & 0xffffu));
                    v1 += bf16_to_f32((unsigned short)(old[0] >> 16));
                    v2 += bf16_to_f32((unsigned short)(old[1] & 0xffffu));
                    v3 += bf16_to_f32((unsigned short)(old[1] >> 16));
                }
                u32x2 ov;
                ov[0] = pack_bf16x2(v0, v1);
                ov[1] = pack_bf16x2(v2, v3);
                *dst = ov;
            }
        }
    }
}

// shapes the kernel takes (host): a short key sequence, enough query blocks to keep every CU busy for several units
// (V^T is read up to column 511: Lk = 512, or a ragged Lk in (448, 512) whose caller pads V^T with finite values to the whole 64-key tile —
// YUME_ATTN_KV_PADDED)
inline bool fits(int64_t Lq, int64_t Lk, int64_t ldvt, bool kv_padded) {
    return Lk > 448 && Lk <= LKMAX && (Lk == LKMAX || kv_padded) && Lq >= 1024 && ldvt >= LKMAX;
}
// MEASURED (profiles/r6_bench_ab_cross_attention_rk.log, the bench's cross-attention launch, two boxes): 115.0 us (first build: P^T fragment
// reads serialised through one register quad, one chain of 32 dependent O MFMAs, 8-byte V^T loads) -> 110.8 us (ring of 8 fragments, four
// O accumulators, permuted K rows) against 89.7 / 97.4 us of the 4-wave streaming kernel on the same boxes. No K / V^T traffic, no item
// boundary, no partial merge — and still slower: with ONE wave per SIMD the unit is a serial chain (32 S MFMAs, row maximum, barrier, 64
// exponentials + pack + 8 KiB of LDS writes, barrier, 32 O MFMAs, store) in which the ~500 VALU instructions and the two barriers run beside
// an idle matrix pipe: ~8000 clocks per unit against 2048 of MFMA. The streaming kernel hides the same arithmetic behind its second
// workgroup on the CU. What this design still needs is the interleave attn_fwd7 writes by hand (the softmax of unit u inside the MFMA
// gaps of S(u + 1) / O(u - 1)); as built it stays OPT-IN: variant 9, or YUME_ATTN_RK=1 for variant 0.
inline bool applies(int64_t Lq, int64_t Lk, int64_t ldvt, bool kv_padded) {
    static const bool on = [] { const char* v = getenv("YUME_ATTN_RK"); return v && atoi(v) != 0; }();
    return on && fits(Lq, Lk, ldvt, kv_padded);
}

inline void launch(const AttnArgs& a, int ncu, hipStream_t st) {
    AttnArgs b = a;
    b.q_lo = 0;
    const int nqb = (int)((b.Lq + QB - 1) / QB);
    const int nunit = nqb * b.H;
    const int g = nunit < ncu ? nunit : ncu;
    if (b.Lk < LKMAX) hipLaunchKernelGGL(attn_cross_rk_kernel<true>, dim3((unsigned)g), dim3(256), 0, st, b, nqb, nunit);
    else hipLaunchKernelGGL(attn_cross_rk_kernel<false>, dim3((unsigned)g), dim3(256), 0, st, b, nqb, nunit);
}

}  // namespace attn_rk
