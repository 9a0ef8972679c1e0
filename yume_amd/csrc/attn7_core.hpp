// attn7_core.hpp — the phase machinery of the one-wave-per-SIMD attention kernels, shared by attn_fwd7.hip (one workgroup per query block)
// and attn_fwd8.hip (persistent workgroups, one continuous K / V^T stream across their items). See attn_fwd7.hip for the design notes.
// Everything lives in an anonymous namespace: each including .hip file gets its own copy.
#pragma once
#include "common.hpp"
#include "attn_args.hpp"
#include "trace.hpp"
#include <type_traits>

namespace {

constexpr int KT = 64;
constexpr int D = 128;
constexpr int SLOT = KT * D * 2;          // 16 KiB per tile image
constexpr int NS = 4;                     // slots per operand
constexpr int VB = NS * SLOT;             // V^T slots start here
constexpr int LDS7 = 2 * NS * SLOT;       // 128 KiB
constexpr int QB7 = 256;                  // queries per workgroup
#ifndef A7_RD
#define A7_RD 4
#endif
// r6 ("swap-free P layout"): the K tile is staged with its rows PERMUTED inside every 32-key block — LDS row m holds key m with
// bits 2 and 3 exchanged (pure address arithmetic in the loop-invariant LDS-DMA source offsets) — so that a lane's score accumulator holds,
// per 16-key step, the EIGHT CONSECUTIVE keys 8 (lane >> 5) + 0..7 instead of 4 (lane >> 5) + (0..3) and 8 + 4 (lane >> 5) + (0..3): its
// exponentials pack straight into the P^T B fragment the V^T image expects. The 8 v_permlane32_swap per block and tile (and the two idle
// states each needs behind the cvt_pk that feeds it) are gone; the V^T image the QKV epilogue writes is untouched. 0 = the r2-r5 layout (A/B).
#ifndef A7_SWAPFREE
#define A7_SWAPFREE 1
#endif
// first key (inside a tile) of a lane's accumulator element r of block b, and of the lane half
#if A7_SWAPFREE
#define A7_KEY_OF(b, r) (32 * (b) + 16 * ((r) >> 3) + ((r) & 7))
#define A7_KEYH_SHIFT 3
#else
#define A7_KEY_OF(b, r) (32 * (b) + ((r) & 3) + 8 * ((r) >> 2))
#define A7_KEYH_SHIFT 2
#endif
constexpr int RD = A7_RD;                     // V^T fragment ring depth: a fragment is read RD MFMA gaps before its MFMA (8 measured the same)
constexpr float NEG_BIG = -1.0e30f;
constexpr float OVERFLOW_LOG2 = 13.0f;       // deferred rescale: exponentials stay below 2^13 against the running base
constexpr float OVERFLOW_SUM = 8192.0f;       // = 2^13: a larger partial row sum (32 exponentials) proves one of them exceeded 2^8

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// ---- the AGPRs this kernel owns -----------------------------------------------------------------------------------------
//   block A: O^T a[0:63] (d block db at 16*db), Q^T a[128:159] (k-step ks at 4*ks);  block B: O^T a[64:127], Q^T a[160:191];
//   a[192:255]: the 16 K fragments of the current key tile (fragment f at 4*f), read from LDS once and used by both blocks
constexpr int OA = 0, OB = 64, QA = 128, QB = 160, KC0 = 192;
#define AG8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define OWNED_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", AG8(1), AG8(2), AG8(3), AG8(4), AG8(5), AG8(6), AG8(7), \
    AG8(8), AG8(9), AG8(10), AG8(11), AG8(12), AG8(13), AG8(14), AG8(15), AG8(16), AG8(17), AG8(18), AG8(19), AG8(20), AG8(21), AG8(22), \
    AG8(23), AG8(24), "a250", "a251", "a252", "a253", "a254", "a255"
// S = K Q^T: D in VGPRs (the softmax reads it), A = K fragment (VGPR, from LDS), B = Q^T fragment a[q:q+3]
// Every MFMA statement also names the softmax state of the OTHER block (Y) as input operands it does not use: that
// pins the VALU work written in the preceding gap to that gap (LLVM otherwise sinks whatever is only needed after the
// phase-end redo branch out from under the MFMAs) without separate statements — an empty asm right behind a VALU write
// costs an s_nop each time.
#define YPINS(y) "v"(y.z.x), "v"(y.z.p[0]), "v"(y.z.p[1]), "v"(y.z.p[2]), "v"(y.z.p[3]), "v"(y.z.p[4]), "v"(y.z.p[5]), "v"(y.z.p[6]), "v"(y.z.p[7]), \
    "v"(y.z.sum0), "v"(y.z.sum1), "v"(y.z.ev), "v"(y.z.od), "v"(y.z.w0), "v"(y.z.w1)
#define MFMA_S0(d, a, q, y) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(d) : "v"(a), "n"(q), "n"((q) + 3), YPINS(y) : "memory", OWNED_AGPRS)
#define MFMA_S(d, a, q, y) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "n"(q), "n"((q) + 3), YPINS(y) : "memory", OWNED_AGPRS)
// O += V^T P^T: C/D = a[o:o+15], A = V^T fragment (VGPR, from LDS), B = P^T fragment (VGPR)
#define MFMA_O(o, a, b, y) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(o), "n"((o) + 15), YPINS(y) : "memory", OWNED_AGPRS)
#define PIN_BLK(y) asm volatile("" ::YPINS(y))
// the same S MFMAs with the K fragment taken from the cache a[k:k+3]
#define MFMA_S0_KC(d, k, q, y) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=&v"(d) : "n"(k), "n"((k) + 3), "n"(q), "n"((q) + 3), YPINS(y) : "memory", OWNED_AGPRS)
#define MFMA_S_KC(d, k, q, y) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(d) : "n"(k), "n"((k) + 3), "n"(q), "n"((q) + 3), YPINS(y) : "memory", OWNED_AGPRS)
// ... carrying one LDS-DMA piece (64 lanes x 16 bytes from sbase + voff[lane] to LDS bytes [lbase + loff, + 1024)): M0 is written in front
// of the MFMA, which is the wait state the LDS-DMA instruction needs after an M0 write (a stand-alone piece pays an s_nop for it), and as
// lbase (SGPR: LDS address of this wave's 1 KiB lane of the slots) + a literal, so that the 32 destinations of a four-tile trip do not
// occupy 32 SGPRs across the loop
#define MFMA_S0_KC_DMA(d, k, q, y, voff, sbase, lbase, loff) asm volatile("s_add_u32 m0, %7, %8\n\tv_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0\n\tglobal_load_lds_dwordx4 %5, %6" \
    : "=&v"(d) : "n"(k), "n"((k) + 3), "n"(q), "n"((q) + 3), "v"(voff), "s"(sbase), "s"(lbase), "n"(loff), YPINS(y) : "memory", "scc", OWNED_AGPRS)
#define MFMA_S_KC_DMA(d, k, q, y, voff, sbase, lbase, loff) asm volatile("s_add_u32 m0, %7, %8\n\tv_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0\n\tglobal_load_lds_dwordx4 %5, %6" \
    : "+v"(d) : "n"(k), "n"((k) + 3), "n"(q), "n"((q) + 3), "v"(voff), "s"(sbase), "s"(lbase), "n"(loff), YPINS(y) : "memory", "scc", OWNED_AGPRS)
// O += V^T P^T whose statement also names the NEXT gap's V^T fragment: the compiler's wait in front of it then covers both (LDS
// returns in order) and the next gap needs none
#define MFMA_O2(o, a, b, a_next, y) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(o), "n"((o) + 15), "v"(a_next), YPINS(y) : "memory", OWNED_AGPRS)
// K fragment F (k-step F>>1, key half F&1) of the K slot at LDS byte offset kb -> a[KC0 + 4F ..]. Untracked by hipcc's s_waitcnt
// bookkeeping: LDS returns in order, and every such load is followed by V^T fragment reads hipcc does wait for before the
// phase ends (or by an explicit s_waitcnt lgkmcnt(0) where it is not), so the data is there when the next phase's MFMAs read it.
#define LOAD_KC(F, addr, off) asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "n"(KC0 + 4 * (F)), "n"(KC0 + 4 * (F) + 3), "n"(off) : "memory", OWNED_AGPRS)
#define LOAD_KC_DYN(F, addr, off) LOAD_KC(F, addr, off)
template <int R>
__device__ __forceinline__ void agpr_set(unsigned v) {
    asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(v), "n"(R) : OWNED_AGPRS);
}
template <int R>
__device__ __forceinline__ float agpr_get() {
    float r;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(r) : "n"(R) : OWNED_AGPRS);
    return r;
}
template <int R>
__device__ __forceinline__ void agpr_scale(float f) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\tv_mul_f32 %0, %0, %1\n\ts_nop 1\n\tv_accvgpr_write_b32 a[%c2], %0" : "=&v"(t) : "v"(f), "n"(R) : OWNED_AGPRS);
}
// compile-time loops over register numbers
template <int R0, int N, typename F>
__device__ __forceinline__ void for_regs(F&& f) {
    if constexpr (N > 0) {
        f(std::integral_constant<int, R0>{});
        for_regs<R0 + 1, N - 1>(f);
    }
}
#define NOP_PAD() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")
// keeps a value (and the instructions that produce it) in the gap it was written in: LLVM otherwise sinks work whose result is only
// needed after the redo branch into the block behind the phase, out from under the MFMAs
#define PIN(x) asm volatile("" : "+v"(x))

__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max2f(float a, float b) {      // no canonicalising v_max in front (the inputs are never sNaN)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return max2f(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ---- one block of 32 queries ------------------------------------------------------------------------------------------
struct Soft {
    float x;            // shifted score of the element whose exponential comes next
    float p[8];         // exponentials not yet packed
    float sum0, sum1;   // partial row sums of the current tile
    unsigned ev, od;    // packed pair waiting for its cross-half swap
    unsigned w0, w1;    // the two P^T words the last swap produced (copies of what went into pf; only there to be pinned)
    float m_run;        // exponent base in use (log2 domain)
    float l_run;        // row sum over this lane's keys
    int need;           // wave-uniform: the tile has to be redone against a new base
};
struct Blk {             // the compiler-managed part of a block (O^T and Q^T are in the owned AGPRs)
    f32x16 s[2];        // S^T of one key tile
    u32x4 pf[4];        // P^T fragments of one key tile
    Soft z;
};

// Piece I of the pipelined softmax of one key tile (32 scores per lane, element e = 16*b + r):
//   I = e      : x_e = s_e * c - m                          (I = 0..31)
//   I = e + 1  : p_e = exp2(x_e)  [masked]                   (I = 1..32)
//   I = e + 2  : row sum                                     (I = 2..33)
//   I = 8g + 9 .. 8g + 11 : pack the 8 exponentials of group g into the P^T fragment g: cvt_pk pair 0 | swap 0, cvt_pk pair 1 |
//                swap 1 (a v_permlane32_swap right behind the cvt_pk that feeds it needs two idle states)
//   I = 31     : wave vote: did any partial row sum leave the range the running base guarantees? (no row maximum on this path)
//   I = 35     : l += sums
// Pieces 0..31 sit in the 32 MFMA gaps of the phase that computes the OTHER block; 32..35 ("drain") sit in the first four
// gaps of the next phase. Re-running pieces 0..31 rebuilds exactly the state the drain expects (the redo path).
// FAST (the caller's Q carries softmax scale * log2(e); no base at all): the scores are the exponents, p = exp2(s) — the shift piece and
// the range vote do not exist. Floating point is scale-invariant, so nothing is lost as long as the row sums stay inside the fp32 range;
// the kernel checks that once per query block at the end and reruns the workgroup on the robust pieces if it does not hold.
template <int I, bool MASK, bool FAST>
__device__ __forceinline__ void sm_piece(Soft& z, const f32x16 (&s)[2], u32x4 (&pf)[4], float c, int keyb, int Lk) {
    auto row_sum = [&]() {
#if A7_DOT2_ROWSUM
        if constexpr (FAST) return;            // (experiment: the pack pieces sum the bf16 pairs they have just made, v_dot2c_f32_bf16)
#endif
        if constexpr (I >= 2 && I <= 33) {
            constexpr int e = I - 2;
            if constexpr (e == 0) z.sum0 = z.p[0];
            else if constexpr (e == 1) z.sum1 = z.p[1];
            else if constexpr ((e & 1) != 0) z.sum1 += z.p[e & 7];
            else z.sum0 += z.p[e & 7];
        }
    };
    auto pack = [&]() {
        if constexpr (I >= 9) {
            constexpr int g = (I - 9) >> 3, k = (I - 9) & 7;
#if A7_SWAPFREE
            // the lane's 8 exponentials of the group ARE its 8 consecutive keys of the k-step: words (p0,p1) (p2,p3) (p4,p5) (p6,p7)
            if constexpr (k == 0 || k == 1) {          // cvt_pk of the pairs k and 2 + k (k = 0 reads p0, p1, p4, p5; k = 1 reads p2, p3, p6, p7)
                z.ev = pack_bf16x2(z.p[2 * k], z.p[2 * k + 1]);
                z.od = pack_bf16x2(z.p[4 + 2 * k], z.p[4 + 2 * k + 1]);
                z.w0 = z.ev;
                z.w1 = z.od;
                pf[g][k] = z.ev;
                pf[g][2 + k] = z.od;
#if A7_DOT2_ROWSUM
                if constexpr (FAST) {
                    // row sum on the dot unit: the two bf16 pairs just packed times (1, 1) — 4 issues per 8 exponentials instead of 8 adds; the sum is
                    // then the sum of the ROUNDED exponentials (the ones the P.V product uses), not of the fp32 ones the reference sums
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
                    const bf16x2_t one2 = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
                    if constexpr (g == 0 && k == 0) {
                        z.sum0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, z.ev), one2, 0.f, false);
                        z.sum1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, z.od), one2, 0.f, false);
                    } else {
                        z.sum0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, z.ev), one2, z.sum0, false);
                        z.sum1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, z.od), one2, z.sum1, false);
                    }
                }
#endif
            }
#else
            if constexpr (k == 1 || k == 2) {          // swap of pair k - 1
                const auto r = __builtin_amdgcn_permlane32_swap(z.ev, z.od, false, false);
                z.w0 = r[0];
                z.w1 = r[1];
                pf[g][k - 1] = z.w0;
                pf[g][2 + k - 1] = z.w1;
                if constexpr (k == 2) { z.ev = z.w0; z.od = z.w1; }   // nothing stale stays live as a pin (the swap works in place)
            }
            if constexpr (k == 0 || k == 1) {          // cvt_pk of pair k
                z.ev = pack_bf16x2(z.p[2 * k], z.p[2 * k + 1]);
                z.od = pack_bf16x2(z.p[4 + 2 * k], z.p[4 + 2 * k + 1]);
            }
#endif
        }
    };
    auto expo = [&]() {
        if constexpr (I >= 1 && I <= 32) {
            constexpr int e = I - 1;
#ifdef A7_PIN_MASKED_EXP
            // (attn_fwd8.hip) In the base-free body an exponential depends on nothing but its score, which is complete when the phase starts:
            // with the mask's select behind it, instruction selection emits all 32 of a masked phase up front (hipcc's scheduling barriers
            // only hold what is already in place) and the register file overflows into scratch — reloads that wait, vmcnt(0), for the
            // tile's LDS-DMA pieces. Taking the score through an empty volatile statement ties each one to its own gap.
            float xin = FAST ? s[e >> 4][e & 15] : z.x;
            if constexpr (MASK && FAST) asm volatile("" : "+v"(xin));
            float pv = __builtin_amdgcn_exp2f(xin);
#else
            float pv = __builtin_amdgcn_exp2f(FAST ? s[e >> 4][e & 15] : z.x);
#endif
            if constexpr (MASK) {
                constexpr int b = e >> 4, r = e & 15;
                const int key = keyb + A7_KEY_OF(b, r);
                pv = key < Lk ? pv : 0.f;
            }
            z.p[e & 7] = pv;
        }
    };
    if constexpr (FAST) {
        // without the shift piece the gap's first VALU follows the MFMA statement directly, and hipcc pads whatever it cannot see through
        // an asm statement: an exponential's result read by the first instruction behind it (trans -> VALU forwarding), a cvt_pk's by a
        // swap. So: the exponential first whenever the slot it writes is not an input of this piece's cvt_pk (only the k = 0 piece of a
        // group reads p[0]), the swap never first, and the row sum — whose exponential sat in the middle of the previous gap — last.
        constexpr bool exp_first = I >= 9 && (((I - 9) & 7) == 1 || ((I - 9) & 7) == 2);
        if constexpr (exp_first) expo();
        pack();
        if constexpr (!exp_first) expo();
        row_sum();
    } else {
        row_sum();
        pack();
        expo();
        if constexpr (I <= 31) z.x = __builtin_fmaf(s[I >> 4][I & 15], c, -z.m_run);
        if constexpr (I == 31) {
            // Did the base hold? The exponentials of a tile whose scores exceed the running base by more than 2^13 make the row sum
            // (or the two values not yet summed) exceed 2^13 — including inf on the first tile, whose base is -1e30. No row maximum
            // is reduced on this path; the redo path computes it from S, which stays intact until the next phase.
            const float big = max3f(z.sum0, z.sum1, z.p[30 & 7]);
            z.need = __any((big > OVERFLOW_SUM) | (z.x > OVERFLOW_LOG2));
        }
    }
    if constexpr (I == 35) z.l_run += z.sum0 + z.sum1;
}

#define REP32(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) \
                 M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)

// the tile of block y (O^T at a[YO:YO+63]) has to be redone against a new base (the row maximum, taken here): rescale what was accumulated, rebuild the softmax state
template <int YO, bool MASK>
__device__ __forceinline__ void redo_tile(Blk& y, float c, int keyb, int Lk) {
    NOP_PAD();                                          // pending MFMA results -> accvgpr reads
    float mx = y.s[0][0];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, y.s[b][r]);       // rows >= Lk of a ragged K tile are copies of a valid key: no mask needed
    mx = xhalf_max(mx);
    const float m_new = fmaxf(y.z.m_run, mx * c);
    const float alpha = __builtin_amdgcn_exp2f(y.z.m_run - m_new);
    y.z.m_run = m_new;
    y.z.l_run *= alpha;
    for_regs<YO, 64>([&](auto r) { agpr_scale<decltype(r)::value>(alpha); });
#define YUME_P(i) sm_piece<i, MASK, false>(y.z, y.s, y.pf, c, keyb, Lk);
    REP32(YUME_P)
#undef YUME_P
    NOP_PAD();                                          // accvgpr writes -> MFMA C operands
}

// ---- LDS-DMA ------------------------------------------------------------------------------------------------------------
struct Dma7 {
    const char* kbase;     // K + h*D (bytes), uniform
    const char* vbase;     // V^T + h*D rows (bytes), uniform
    int64_t krow;          // bytes per K row
    unsigned kof[4];       // per-lane source byte offsets of the 4 K pieces of a tile (row rr*16 + tid/16, swizzled chunk)
    unsigned vof[4];       // ... of the 4 V^T pieces (row rr*32 + tid/8, swizzled chunk)
    int kr;                // K row of piece 0
    int kch;               // swizzled K chunk byte offset
    int vc;                // logical V^T chunk (8 keys)
};

__device__ __forceinline__ void dma7_init(Dma7& d, const AttnArgs& p, int h, int tid) {
    d.krow = p.ldk * 2;
    d.kbase = reinterpret_cast<const char*>(p.K + h * D);
    d.vbase = reinterpret_cast<const char*>(p.Vt + (int64_t)h * D * p.ldvt);
    d.kr = tid >> 4;
    d.kch = ((tid & 15) ^ (d.kr & 15)) << 4;              // (the swizzle follows the LDS row)
#if A7_SWAPFREE
    d.kr = (d.kr & 3) | ((d.kr & 4) << 1) | ((d.kr & 8) >> 1);   // ... the SOURCE row is the LDS row with bits 2 and 3 exchanged (stays inside its 16-row piece)
#endif
    const int dd = tid >> 3;
    d.vc = (tid & 7) ^ ((dd >> 1) & 7);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        d.kof[rr] = (unsigned)((d.kr + 16 * rr) * d.krow) + d.kch;
        d.vof[rr] = (unsigned)((dd + 32 * rr) * p.ldvt * 2) + (d.vc << 4);    // (dd + 32 rr) >> 1 & 7 == dd >> 1 & 7
    }
}

// One LDS-DMA piece: 64 lanes x 16 bytes from sbase + voff[lane] to LDS bytes [lds_dst, lds_dst + 1024). Inline asm so that
// the address is SGPR base + 32-bit VGPR offset (hipcc builds 64-bit per-lane pointers for the builtin: 16 more VGPRs and a
// 64-bit add per piece) and so that hipcc does not order later LDS reads behind it with vmcnt(0); completion is counted by
// the explicit s_waitcnt vmcnt(N) + barrier of the tile loop. M0 is written in the statement that uses it.
__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// whole tiles, any tile (the ragged last one clamps its sources; fix7_v zeroes what must be zero afterwards)
__device__ __forceinline__ void dma7_k(const Dma7& d, const AttnArgs& p, int t, bool last_ragged, unsigned slot, int wave) {
    const char* base = d.kbase + (int64_t)t * KT * d.krow;
    const unsigned l = slot + wave * 1024;
    if (!last_ragged) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) glds16(base, d.kof[rr], l + rr * 4096);
    } else {
        int nrow = p.Lk - t * KT;             // 1..63 valid rows; the others are copies of the last one (their P is masked)
        asm volatile("" : "+s"(nrow));        // rare path: keep its address arithmetic here instead of hoisted (and spilled) in front of the tile loop
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = d.kr + 16 * rr;
            const int rc = r < nrow ? r : nrow - 1;
            glds16(base, (unsigned)rc * (unsigned)d.krow + (unsigned)d.kch, l + rr * 4096);
        }
    }
}
__device__ __forceinline__ void dma7_v(const Dma7& d, const AttnArgs& p, int t, bool last_ragged, unsigned slot, int wave) {
    const char* base = d.vbase + (int64_t)t * KT * 2;
    const unsigned l = slot + wave * 1024;
    unsigned back = 0;
    if (last_ragged) {
        int kc = t * KT + d.vc * 8;
        asm volatile("" : "+v"(kc));          // rare path: not hoisted
        const int kmax = (int)p.ldvt - 8;
        if (kc > kmax) back = (unsigned)((kc - kmax) * 2);         // stay inside the row; such a chunk is zeroed afterwards
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) glds16(base, d.vof[rr] - back, l + rr * 4096);
}
// keys >= Lk of the ragged last V^T tile -> 0 (0 * stale bits must be 0), by the thread whose DMA brought the chunk
__device__ __forceinline__ void fix7_v(const Dma7& d, const AttnArgs& p, int t, char* slot, int tid) {
    int nvalid = p.Lk - (t * KT + d.vc * 8);
    asm volatile("" : "+v"(nvalid));          // rare path (once per workgroup): its lane masks are not worth registers across the tile loop
    if (nvalid >= 8) return;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        u32x4* c = reinterpret_cast<u32x4*>(slot + rr * 4096 + tid * 16);
        u32x4 x = *c;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (2 * w >= nvalid) x[w] = 0u;
            else if (2 * w + 1 >= nvalid) x[w] &= 0xffffu;
        }
        *c = x;
    }
}

// timing experiments (wrong results): the V^T fragment reads of block B's phases / the K cache refills left out
#ifndef ATTN_ABLATE_VB
#define ATTN_ABLATE_VB 0
#endif
// timing ablations of the steady tile's meeting point (WRONG results; profiles/r6_attention_steady_barrier_ablation.log): which of the
// counted wait and the barrier the waves are parked at
// experiment (second candidate): the row sums of the base-free body on v_dot2_f32_bf16 (profiles/r6_attention_dot2_rowsum.log)
#ifndef A7_DOT2_ROWSUM
#define A7_DOT2_ROWSUM 0
#endif
#ifndef ATTN_ABLATE_STEADY_BAR
#define ATTN_ABLATE_STEADY_BAR 0
#endif
#ifndef ATTN_ABLATE_STEADY_WAIT
#define ATTN_ABLATE_STEADY_WAIT 0
#endif
#ifndef ATTN_ABLATE_KC
#define ATTN_ABLATE_KC 0
#endif
// ---- a phase: 32 MFMA gaps ------------------------------------------------------------------------------------------------
struct Ctx {
    char* smem;
    unsigned lds0;             // LDS byte address of smem
    unsigned lbase;            // lds0 + wave * 1024: this wave's 1 KiB lane of every 4 KiB quarter slot (LDS-DMA destinations)
    int koff[8], voff[4];      // per-lane fragment offsets inside a K slot / inside V^T slot 0 (VB included)
    float c;                   // softmax scale * log2(e)
    int keyh;                  // 4 * (lane >> 5)
    int Lk;
    int wave;
};

// V^T fragment F - 16 (key group (F-16)>>2, d block (F-16)&3) of the V^T slot at byte offset vb (F = 16..31: the MFMA gap that uses it)
template <int F>
__device__ __forceinline__ u32x4 frag(const Ctx& cx, int vb) {
    static_assert(F >= 16 && F < 32, "V^T fragments belong to gaps 16..31");
    return *reinterpret_cast<const u32x4*>(cx.smem + vb + cx.voff[(F - 16) >> 2] + ((F - 16) & 3) * (32 * 128));
}

// all 16 K fragments of the K slot at byte offset kb -> the AGPR cache, and wait for them (prologue / before the steady loop)
__device__ __forceinline__ void fill_kcache(const Ctx& cx, int kb) {
#define YUME_KC(f) LOAD_KC_DYN(f, cx.koff[(f) >> 1] + kb, ((f) & 1) ? 32 * 256 : 0);
    YUME_KC(0) YUME_KC(1) YUME_KC(2) YUME_KC(3) YUME_KC(4) YUME_KC(5) YUME_KC(6) YUME_KC(7)
    YUME_KC(8) YUME_KC(9) YUME_KC(10) YUME_KC(11) YUME_KC(12) YUME_KC(13) YUME_KC(14) YUME_KC(15)
#undef YUME_KC
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// X (O^T at a[XO..], Q^T at a[XQ..]): the block whose MFMAs run — S of the K tile in the AGPR cache if DO_S, O += V^T P of the tile in
//    V^T slot vb if DO_PV — and whose previous softmax drains in the first gaps (DRAIN, tile starting at key jx);
// Y (O^T at a[YO..]): the block whose softmax pieces fill the gaps (SM, tile starting at key jy).
// NKB >= 0: refill the K cache with the K slot at byte offset NKB (compile-time: it folds into the ds_read offsets), fragment f in
//   gap f + 1, right after its last reader in this phase.
// DMA: gaps 0..7 issue one LDS-DMA piece each (K tile at kg -> kslot, V^T tile at vg -> vslot).
// DK / DV (DMA): byte offsets of the destination K / V^T slots from smem.
template <bool FAST, int XO, int XQ, int YO, bool DO_S, bool DO_PV, bool DRAIN, bool MASKX, bool SM, bool MASKY, bool DMA, int NKB = -1, int DK = 0, int DV = 0>
__device__ __forceinline__ void phase(const Ctx& cx, Blk& X, Blk& Y, u32x4 (&ring)[RD], int vb, int jx, int jy,
                                      const Dma7& dp, const char* kg, const char* vg) {
    static_assert(!DMA || DO_S, "the LDS-DMA pieces ride on the score MFMAs of gaps 0..7");
    __builtin_amdgcn_sched_barrier(0);
#define YUME_DV(i) ((i) < 4 ? dp.kof[(i) & 3] : dp.vof[(i) & 3])
#define YUME_DS(i) ((i) < 4 ? kg : vg)
#define YUME_DL(i) (((i) < 4 ? DK : DV) + ((i) & 3) * 4096)
#define YUME_GAP(i)                                                                                          \
    {                                                                                                        \
        if constexpr (SM && ((i) < 16 ? !DO_S : !DO_PV)) PIN_BLK(Y);                                         \
        if constexpr ((i) < 16) {                                                                            \
            if constexpr (DO_S) {                                                                            \
                if constexpr (((i) >> 1) == 0) {                                                             \
                    if constexpr (DMA) MFMA_S0_KC_DMA(X.s[(i) & 1], KC0 + 4 * ((i) & 15), XQ, Y, YUME_DV((i) & 7), YUME_DS((i) & 7), cx.lbase, YUME_DL((i) & 7)); \
                    else MFMA_S0_KC(X.s[(i) & 1], KC0 + 4 * ((i) & 15), XQ, Y);                              \
                } else if constexpr (DMA && (i) < 8) {                                                       \
                    MFMA_S_KC_DMA(X.s[(i) & 1], KC0 + 4 * ((i) & 15), XQ + 4 * ((i) >> 1), Y, YUME_DV((i) & 7), YUME_DS((i) & 7), cx.lbase, YUME_DL((i) & 7)); \
                } else {                                                                                     \
                    MFMA_S_KC(X.s[(i) & 1], KC0 + 4 * ((i) & 15), XQ + 4 * ((i) >> 1), Y);                   \
                }                                                                                            \
            }                                                                                                \
        } else if constexpr (DO_PV) {                                                                        \
            if constexpr (((i) & 1) == 0) MFMA_O2(XO + 16 * (((i) - 16) & 3), ring[(i) & (RD - 1)], X.pf[((i) - 16) >> 2], ring[((i) + 1) & (RD - 1)], Y); \
            else MFMA_O(XO + 16 * (((i) - 16) & 3), ring[(i) & (RD - 1)], X.pf[((i) - 16) >> 2], Y);          \
        }                                                                                                    \
        if constexpr (DO_PV && (i) + RD >= 16 && (i) + RD < 32 && !(ATTN_ABLATE_VB && XO == OB)) ring[(i) & (RD - 1)] = frag<(((i) + RD) & 15) + 16>(cx, vb); \
        if constexpr (NKB >= 0 && (i) >= 1 && (i) <= 16 && !ATTN_ABLATE_KC)                    \
            LOAD_KC(((i) - 1) & 15, cx.koff[(((i) - 1) & 15) >> 1], (NKB < 0 ? 0 : NKB) + ((((i) - 1) & 1) ? 32 * 256 : 0)); \
        if constexpr (DRAIN && (i) >= 1 && (i) < 5) PIN_BLK(X);     /* drain piece of the previous gap stays there */ \
        if constexpr (DRAIN && (i) < 4) sm_piece<32 + ((i) & 3), MASKX, FAST>(X.z, X.s, X.pf, cx.c, jx + cx.keyh, cx.Lk); \
        if constexpr (SM) sm_piece<(i), MASKY, FAST>(Y.z, Y.s, Y.pf, cx.c, jy + cx.keyh, cx.Lk);     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
    REP32(YUME_GAP)
#undef YUME_GAP
#undef YUME_DV
#undef YUME_DS
#undef YUME_DL
    if constexpr (NKB >= 0 && !DO_PV) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // no V^T reads behind the K loads to order them
    if constexpr (SM) {
        PIN_BLK(Y);
        if constexpr (!FAST)
            if (__builtin_expect(Y.z.need, 0)) redo_tile<YO, MASKY>(Y, cx.c, jy + cx.keyh, cx.Lk);
    }
}

template <int XO>
__device__ __forceinline__ void store_block(const AttnArgs& p, const Blk& x, int q, int h, int hi) {
    const float l_tot = xhalf_sum(x.z.l_run);
    const float inv = 1.0f / l_tot;
    float o[64];
    for_regs<0, 64>([&](auto r) { o[decltype(r)::value] = agpr_get<XO + decltype(r)::value>() * inv; });
    if (q < p.Lq) {
        unsigned short* op = p.O + (int64_t)q * p.ldo + h * D + 4 * hi;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = o[16 * db + 4 * g + 0], v1 = o[16 * db + 4 * g + 1];
                float v2 = o[16 * db + 4 * g + 2], v3 = o[16 * db + 4 * g + 3];
                u32x2* dst = reinterpret_cast<u32x2*>(op + 32 * db + 8 * g);
                if (p.accumulate) {
                    const u32x2 old = *dst;
                    v0 += bf16_to_f32((unsigned short)(old[0] & 0xffffu));
                    v1 += bf16_to_f32((unsigned short)(old[0] >> 16));
                    v2 += bf16_to_f32((unsigned short)(old[1] & 0xffffu));
                    v3 += bf16_to_f32((unsigned short)(old[1] >> 16));
                }
                u32x2 w;
                w[0] = pack_bf16x2(v0, v1);
                w[1] = pack_bf16x2(v2, v3);
                *dst = w;
            }
    }
}

// Q^T fragments of the block's query (lane (q, hi) holds Q[q][16*ks + 8*hi .. +7]) -> a[XQ + 4*ks ..]; O^T = 0; softmax state
template <int XO, int XQ, bool FAST, bool RELOAD = true>
__device__ __forceinline__ void load_q(const AttnArgs& p, Blk& x, int q, int h, int hi) {
    if constexpr (RELOAD) {               // (the rerun on the robust pieces finds Q^T where the first pass left it)
        q = q < p.Lq ? q : p.Lq - 1;
        const unsigned short* qp = p.Q + (int64_t)q * p.ldq + h * D + 8 * hi;
        u32x4 qf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(qp + 16 * ks);
        for_regs<0, 32>([&](auto r) { agpr_set<XQ + decltype(r)::value>(qf[decltype(r)::value >> 2][decltype(r)::value & 3]); });
    }
    for_regs<XO, 64>([&](auto r) { agpr_set<decltype(r)::value>(0u); });
    x.z.m_run = FAST ? 0.f : NEG_BIG;
    x.z.l_run = 0.f;
    x.z.need = 0;
    x.z.sum0 = x.z.sum1 = 0.f;
    x.z.x = 0.f;
    x.z.ev = x.z.od = x.z.w0 = x.z.w1 = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) x.z.p[i] = 0.f;
}

// steady-state tile t (TS = t % 4): 1 <= t, t + 4 < number of FULL tiles; every LDS address is a compile-time constant
// kg / vg: DMA sources of K(t+4) / V^T(t+3), advanced by one tile here (two SALU adds each instead of a 64-bit product per tile)
template <int TS, bool FAST>
__device__ __forceinline__ void steady7(const Ctx& cx, const Dma7& dp, const char*& kg, const char*& vg, int64_t kstep, Blk& A, Blk& B, u32x4 (&ring)[RD]) {
    constexpr int vb = TS * SLOT, nkb = ((TS + 2) & 3) * SLOT;      // V^T(t); K(t+2) for the cache refill (K(t+1) is in the cache)
    constexpr int dk = TS * SLOT, dv = VB + ((TS + 3) & 3) * SLOT;      // K(t+4) takes K(t)'s slot, V^T(t+3) the slot V^T(t-1) left
#if !ATTN_ABLATE_STEADY_WAIT
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // everything older than the previous tile's 8 pieces has landed
#endif
#if !ATTN_ABLATE_STEADY_BAR
    __builtin_amdgcn_s_barrier();
#endif
    phase<FAST, OA, QA, OB, true, true, true, false, true, false, true, -1, dk, dv>(cx, A, B, ring, vb, 0, 0, dp, kg, vg);
    phase<FAST, OB, QB, OA, true, true, true, false, true, false, false, nkb>(cx, B, A, ring, vb, 0, 0, dp, kg, vg);
    kg += kstep;
    vg += KT * 2;
}

// any tile t of the range [.., t1) (runtime slots; the softmax pieces always carry the key mask). nt / ragged describe the whole key
// sequence: only its last tile can be ragged.
// s_waitcnt vmcnt(n), n a multiple of 4 up to 16 (wave-uniform)
__device__ __forceinline__ void wait_vm(int n) {
    if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// any tile t of the range [.., t1) (runtime slots). nt / ragged describe the whole key sequence: only its last tile can be ragged, and
// only the pieces of tiles t and t + 1 are touched here — MASK = false when neither is that tile.
template <bool FAST, bool MASK>
__device__ __forceinline__ void general7(const Ctx& cx, const Dma7& dp, const AttnArgs& p, int t, int t0, int t1, int nt, bool ragged, int tid,
                                         Blk& A, Blk& B, u32x4 (&ring)[RD]) {
    const int last = nt - 1;
    // This tile reads V^T(t) and K(t+1). LDS-DMA pieces land in issue order, so whatever was issued behind those two may stay in flight:
    // after the prologue (K0 | K1 V0 | K2 V1 | K3 V2) that is K(t0+2), V^T(t0+1), K(t0+3), V^T(t0+2); otherwise what tile t-1 issued,
    // K(t+3) and V^T(t+2) — as far as those tiles exist.
    int keep = t == t0 ? 4 * ((t0 + 2 < t1) + (t0 + 1 < t1) + (t0 + 3 < t1) + (t0 + 2 < t1)) : 4 * ((t + 3 < t1) + (t + 2 < t1));
    // V^T(last) was issued three tiles ago (or in the prologue): zero its keys >= Lk before the barrier that precedes its first read
    const bool fix = ragged && t1 == nt && t == (last - 2 > t0 ? last - 2 : t0);
    if (fix) keep = 0;
    wait_vm(keep);
    if (fix) fix7_v(dp, p, last, cx.smem + VB + (last & 3) * SLOT, tid);
    __builtin_amdgcn_s_barrier();
    if (t + 4 < t1) dma7_k(dp, p, t + 4, ragged && t + 4 == last, cx.lds0 + ((t + 4) & 3) * SLOT, cx.wave);
    if (t + 3 < t1) dma7_v(dp, p, t + 3, ragged && t + 3 == last, cx.lds0 + VB + ((t + 3) & 3) * SLOT, cx.wave);
    const int vb = (t & 3) * SLOT;
    const int j = t * KT;
    if (t + 1 < t1) {
        fill_kcache(cx, ((t + 1) & 3) * SLOT);            // K(t+1), published by the barrier above
        phase<FAST, OA, QA, OB, true, true, true, MASK, true, MASK, false>(cx, A, B, ring, vb, j, j, dp, nullptr, nullptr);
        phase<FAST, OB, QB, OA, true, true, true, MASK, true, MASK, false>(cx, B, A, ring, vb, j, j + KT, dp, nullptr, nullptr);
    } else {
        phase<FAST, OA, QA, OB, false, true, true, MASK, true, MASK, false>(cx, A, B, ring, vb, j, j, dp, nullptr, nullptr);
        phase<FAST, OB, QB, OA, false, true, true, MASK, false, MASK, false>(cx, B, A, ring, vb, j, j, dp, nullptr, nullptr);
    }
}

// unnormalised O^T, running base and row sum of one key range -> the scratch attn_combine_kernel merges (layout: attn_args.hpp)
template <int XO>
__device__ __forceinline__ void store_partial(const AttnArgs& p, const Blk& x, int q, int h, int hi, int sp, int64_t rows, int row0) {
    const float l_part = xhalf_sum(x.z.l_run);
    float o[64];
    for_regs<0, 64>([&](auto r) { o[decltype(r)::value] = agpr_get<XO + decltype(r)::value>(); });
    if (q < p.Lq) {
        const int64_t r = q - row0;
        float* po = p.part_o + ((int64_t)sp * rows + r) * ((int64_t)p.H * D) + h * D + 4 * hi;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(po + 32 * db + 8 * g) = f32x4{o[16 * db + 4 * g + 0], o[16 * db + 4 * g + 1], o[16 * db + 4 * g + 2], o[16 * db + 4 * g + 3]};
        if (hi == 0) {
            float* pm = p.part_ml + (((int64_t)sp * rows + r) * p.H + h) * 2;
            pm[0] = x.z.m_run;
            pm[1] = l_part;
        }
    }
}

// One pass over the key range [t0, t1) for the workgroup's 256 queries: Q^T / O^T / softmax state set up, prologue DMA, the tile loop.
// FAST: see sm_piece. RELOAD = false: Q^T is still in its AGPRs (the robust rerun of a FAST pass).
template <bool FAST, bool RELOAD>
__device__ __forceinline__ void run_keys(const AttnArgs& p, const Ctx& cx, const Dma7& dp, Blk& A, Blk& B, int q0, int ql, int h, int hi,
                                         int t0, int t1, int nt, int tid) {
    const int wave = cx.wave;
    const bool ragged = (p.Lk % KT) != 0;
    const int last = nt - 1;
    const int tsteady = (ragged && t1 == nt) ? t1 - 1 : t1;      // tiles below this index are full AND inside the range
    load_q<OA, QA, FAST, RELOAD>(p, A, q0 + ql, h, hi);
    load_q<OB, QB, FAST, RELOAD>(p, B, q0 + 32 + ql, h, hi);
    __builtin_amdgcn_sched_barrier(0);
    // ---- prologue DMA: K(t0) | K(t0+1) V(t0) | K(t0+2) V(t0+1) | K(t0+3) V(t0+2) ----
    dma7_k(dp, p, t0, ragged && last == t0, cx.lds0 + (t0 & 3) * SLOT, wave);
    if (t0 + 1 < t1) dma7_k(dp, p, t0 + 1, ragged && last == t0 + 1, cx.lds0 + ((t0 + 1) & 3) * SLOT, wave);
    dma7_v(dp, p, t0, ragged && last == t0, cx.lds0 + VB + (t0 & 3) * SLOT, wave);
    if (t0 + 2 < t1) dma7_k(dp, p, t0 + 2, ragged && last == t0 + 2, cx.lds0 + ((t0 + 2) & 3) * SLOT, wave);
    if (t0 + 1 < t1) dma7_v(dp, p, t0 + 1, ragged && last == t0 + 1, cx.lds0 + VB + ((t0 + 1) & 3) * SLOT, wave);
    if (t0 + 3 < t1) dma7_k(dp, p, t0 + 3, ragged && last == t0 + 3, cx.lds0 + ((t0 + 3) & 3) * SLOT, wave);
    if (t0 + 2 < t1) dma7_v(dp, p, t0 + 2, ragged && last == t0 + 2, cx.lds0 + VB + ((t0 + 2) & 3) * SLOT, wave);

    __builtin_amdgcn_sched_barrier(0);
    if (t0 + 3 < t1) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");      // K(t0) has landed (the Q loads are older still)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    NOP_PAD();
    TRACE_STAMP(3);      // (experiment builds, trace.hpp: prologue done / steady loop entered / steady loop left)

    u32x4 ring[RD];
    // S_A(t0), then S_B(t0) beside softmax_A(t0)
    {
        const int j0 = t0 * KT;
        fill_kcache(cx, (t0 & 3) * SLOT);
        phase<FAST, OA, QA, OB, true, false, false, true, false, true, false>(cx, A, B, ring, 0, j0, j0, dp, nullptr, nullptr);
        phase<FAST, OB, QB, OA, true, false, false, true, true, true, false>(cx, B, A, ring, 0, j0, j0, dp, nullptr, nullptr);
    }

    int t = t0;
#pragma unroll 1
    while (t < t1) {
        if (t > t0 && (t & 3) == 1 && t + 7 < tsteady) {
            fill_kcache(cx, 2 * SLOT);          // K(t+1) (slot 2 here), published by the barrier of the previous tile
            // A's score registers are dead here and the loop's first MFMAs overwrite them. Claim them NOW: if hipcc parked a spill
            // reload in them on the way here, its wait for that load lands in front of the loop instead of inside it, where an
            // s_waitcnt vmcnt(0) would also wait for every LDS-DMA piece in flight (tests/test_attn7_isa.py checks the loop).
            asm volatile("" : "=v"(A.s[0]), "=v"(A.s[1]));
            const int64_t kstep = (int64_t)KT * dp.krow;
            const char* kg = dp.kbase + (int64_t)(t + 4) * kstep;
            const char* vg = dp.vbase + (int64_t)(t + 3) * KT * 2;
            TRACE_STAMP(4);
#pragma unroll 1
            for (; t + 7 < tsteady; t += 4) {        // t % 4 == 1; the four calls issue tiles up to t + 7 (all full, all in range)
                steady7<1, FAST>(cx, dp, kg, vg, kstep, A, B, ring);
                steady7<2, FAST>(cx, dp, kg, vg, kstep, A, B, ring);
                steady7<3, FAST>(cx, dp, kg, vg, kstep, A, B, ring);
                steady7<0, FAST>(cx, dp, kg, vg, kstep, A, B, ring);
            }
            TRACE_STAMP(5);
        } else {
            if (ragged && t + 1 >= last) general7<FAST, true>(cx, dp, p, t, t0, t1, nt, ragged, tid, A, B, ring);
            else general7<FAST, false>(cx, dp, p, t, t0, t1, nt, ragged, tid, A, B, ring);
            ++t;
        }
    }
    NOP_PAD();
}

// FAST pass only: did every row of this wave's block stay inside the range? The row sum has to be finite (an overflowing exponential —
// a score above 127 — makes it inf) and not tiny (a row whose scores all sit below about -100 loses its small terms to the flush-to-zero
// of exp2, or sums to 0), and no O^T element may be inf / NaN (0 * x summed over the block is 0 exactly when every x is finite).
// The bounds leave 2^8 of headroom (row sum < 2^120, |O^T| < 2^120: 0 * (2^8 x) is 0 exactly when |x| < 2^120): up to four key-range
// pieces of one row pass this check at the same base (m = 0) and attn_combine_kernel adds their l and O without rescaling.
template <int XO>
__device__ __forceinline__ bool block_in_range(const Blk& x) {
    const float l_tot = xhalf_sum(x.z.l_run);
    float z = 0.f;
    for_regs<0, 64>([&](auto r) { z = __builtin_fmaf(agpr_get<XO + decltype(r)::value>() * 0x1p8f, 0.f, z); });
    return (l_tot > 0x1p-100f) & (l_tot < 0x1p120f) & (z == 0.f);
}

}  // namespace
