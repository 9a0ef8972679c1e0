// gemm_w4.hpp — the dense bf16 GEMM with ONE wave per SIMD (r3):   acc[m,n] = sum_k A[m,k] * W[n,k],   256 x 256 x 64 tiles.
//
// Why another schedule. The 8-wave kernel of gemm_core.hpp (two waves per SIMD, each 128 x 64 of the tile) issues, per 64 MFMAs of a
// wave, 24 fragment reads, 8 LDS-DMA pieces with per-lane 64-bit addresses and 8 workgroup barriers: its matrix pipe is 56-70 % busy
// (profiles/r2_rejected_experiments.md). The vendor library's best kernel for these shapes (rocprofv3: `..._MT256x256x64_MI16x16x1`;
// its code object says 256 threads, 133,120 B of LDS, 249 VGPRs + accumulators) reaches 87 %: four waves, one per SIMD, each owning a
// 128 x 128 quarter = 64 accumulators of 16x16 = all 256 AGPRs. Per K tile a wave then issues 128 MFMAs against 32 fragment reads
// (0.25 per MFMA instead of 0.375), 16 DMA pieces and a few barriers, everything placed one filler per MFMA gap.
// This file is that structure written for this library's operand layout:
//   * 4 waves (2 x 2), wave (wr, wc) owns rows wr*128 + [0,128), columns wc*128 + [0,128) of the tile: accumulator tile (i, j)
//     (16 x 16) lives in a[(i*8+j)*4 .. +3]. The AGPRs are OWNED by this file: named literally in the MFMA statements and listed as
//     clobbers (the convention of attn_fwd7.hip; tests/test_gemm_w4_isa.py audits the compiled loop for it);
//   * LDS: two K-tile buffers of A | B, each operand a row-major [256][64] bf16 image (128-byte rows) with the bank swizzle of
//     gemm_core.hpp applied on the DMA source (16-byte chunk c of row r holds logical chunk c ^ (r & 7)): 128 KiB;
//   * a K tile = two k-steps of 32; fragment sets F0 / F1 (8 A + 8 B fragments each, 128 VGPRs together);
//         segment 1: 64 MFMAs on F0(t); the 16 reads of F1(t) ride in its first gaps; `lgkmcnt(0)` + barrier (every wave is done
//                    with buffer t); the 16 DMA pieces of tile t+2 (into the buffer tile t just left) ride in MFMA statements
//         segment 2: 64 MFMAs on F1(t); counted `vmcnt` (tile t+1 landed: its pieces were issued a whole K tile ago) + barrier;
//                    the 16 reads of F0(t+1)
//     = 2 barriers and 2 waits per 128 MFMAs; no VALU and ~10 SALU inside the loop (source pointers advance by 128 bytes per tile;
//     M0 = SGPR + literal inside the DMA statement; per-lane source offsets are loop-invariant VGPRs, row-clamped at the M / N edge);
//   * epilogues as in gemm_core.hpp: bf16 tiles leave through an LDS image as whole 512-byte rows (the K-major V^T tiles too, as whole
//     rows of 256 tokens), fp32 / residual tiles straight from the accumulators with their loads batched.
// Roofline: MFMA bf16 dense; 2*M*N*K flop per launch.
#pragma once
#include "gemm_core.hpp"
#include "trace.hpp"

namespace gemm_w4 {
using namespace gemm_core;

constexpr int NTHR_W4 = 256;
constexpr int OPER_BYTES = 256 * 128;        // one operand of one K tile
constexpr int BUF_BYTES = 2 * OPER_BYTES;    // A | B
constexpr int LDS_W4 = 2 * BUF_BYTES;        // 128 KiB

#define W4_AG8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define W4_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", W4_AG8(1), W4_AG8(2), W4_AG8(3), W4_AG8(4), W4_AG8(5), W4_AG8(6), \
    W4_AG8(7), W4_AG8(8), W4_AG8(9), W4_AG8(10), W4_AG8(11), W4_AG8(12), W4_AG8(13), W4_AG8(14), W4_AG8(15), W4_AG8(16), W4_AG8(17), W4_AG8(18), \
    W4_AG8(19), W4_AG8(20), W4_AG8(21), W4_AG8(22), W4_AG8(23), W4_AG8(24), "a250", "a251", "a252", "a253", "a254", "a255"

// acc tiles T0, T1 (T = i*8 + j) += X * Y    (X = first MFMA operand: its 16 rows become the 4-consecutive index of the result lane).
// Two MFMAs per statement: hipcc pads every boundary between two asm statements with an s_nop, which would take an issue slot per MFMA gap.
#define W4_MFMA2(T0, X0, Y0, T1, X1, Y1)                                                                                                         \
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c4:%c5], %0, %1, a[%c4:%c5]\n\tv_mfma_f32_16x16x32_bf16 a[%c6:%c7], %2, %3, a[%c6:%c7]"             \
                 ::"v"(X0), "v"(Y0), "v"(X1), "v"(Y1), "n"((T0) * 4), "n"((T0) * 4 + 3), "n"((T1) * 4), "n"((T1) * 4 + 3) : "memory", W4_AGPRS)
// ... carrying one LDS-DMA piece: 64 lanes x 16 bytes from sbase + voff[lane] to LDS bytes [lbase + loff, + 1024). M0 is written in
// front of the first MFMA, which is the wait state the LDS-DMA needs behind an M0 write.
#define W4_MFMA2_DMA(T0, X0, Y0, T1, X1, Y1, voff, sbase, lbase, loff)                                                                            \
    asm volatile("s_add_u32 m0, %10, %11\n\tv_mfma_f32_16x16x32_bf16 a[%c4:%c5], %0, %1, a[%c4:%c5]\n\tglobal_load_lds_dwordx4 %8, %9\n\t"         \
                 "v_mfma_f32_16x16x32_bf16 a[%c6:%c7], %2, %3, a[%c6:%c7]"                                                                       \
                 ::"v"(X0), "v"(Y0), "v"(X1), "v"(Y1), "n"((T0) * 4), "n"((T0) * 4 + 3), "n"((T1) * 4), "n"((T1) * 4 + 3), "v"(voff), "s"(sbase),   \
                 "s"(lbase), "n"(loff) : "memory", "scc", W4_AGPRS)
// ... the same with the source given by a buffer descriptor (conv_w4.hpp): address = srd.base + soff + voff[lane]; a lane whose voff is
// beyond srd.num_records (0xffffffff = "this tap lies in the zero padding") is out of range and the LDS receives zeros for it
#define W4_MFMA2_DMAB(T0, X0, Y0, T1, X1, Y1, voff, srd, soff, lbase, loff)                                                                           \
    asm volatile("s_add_u32 m0, %11, %12\n\tv_mfma_f32_16x16x32_bf16 a[%c4:%c5], %0, %1, a[%c4:%c5]\n\tbuffer_load_dwordx4 %8, %9, %10 offen lds\n\t"  \
                 "v_mfma_f32_16x16x32_bf16 a[%c6:%c7], %2, %3, a[%c6:%c7]"                                                                           \
                 ::"v"(X0), "v"(Y0), "v"(X1), "v"(Y1), "n"((T0) * 4), "n"((T0) * 4 + 3), "n"((T1) * 4), "n"((T1) * 4 + 3), "v"(voff), "s"(srd),         \
                 "s"(soff), "s"(lbase), "n"(loff) : "memory", "scc", W4_AGPRS)
#define W4_DMAB(voff, srd, soff, lbase, loff) \
    asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(lbase), "n"(loff) : "memory", "scc")
// a stand-alone piece (prologue)
#define W4_DMA(voff, sbase, lbase, loff) \
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lbase), "n"(loff) : "memory", "scc")

template <int R>
__device__ __forceinline__ f32x4 acc_tile() {      // accumulator tile R -> VGPRs
    f32x4 v;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "n"(R * 4), "n"(R * 4 + 1), "n"(R * 4 + 2), "n"(R * 4 + 3) : W4_AGPRS);
    return v;
}
template <int R>
__device__ __forceinline__ void acc_tile_add(f32x4 d) {      // accumulator tile R += d
    f32x4 v = acc_tile<R>();
    v += d;
    asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\tv_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
                 ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "n"(R * 4), "n"(R * 4 + 1), "n"(R * 4 + 2), "n"(R * 4 + 3) : W4_AGPRS);
}
template <int R0, int N>
__device__ __forceinline__ void acc_zero() {
    if constexpr (N > 0) {
        asm volatile("v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c1], 0\n\tv_accvgpr_write_b32 a[%c2], 0\n\tv_accvgpr_write_b32 a[%c3], 0"
                     ::"n"(R0), "n"(R0 + 1), "n"(R0 + 2), "n"(R0 + 3) : W4_AGPRS);
        acc_zero<R0 + 4, N - 4>();
    }
}
template <int I0, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, N - 1>(f);
    }
}

struct Ctx {
    char* smem;
    bf16x8_t fa[2][8], fb[2][8];    // fragment sets [k-step][i or j]
    // per-lane LDS byte offsets of fragment 0 (fragment i: + i * 2048): k-step 1 in the buffer of the CURRENT tile, k-step 0 in the buffer of
    // the NEXT tile; all four flip (^ BUF_BYTES) after every tile
    unsigned ra1, rb1, ra0n, rb0n;
    unsigned va[8], vb[8];          // per-lane source byte offsets of this wave's 8 DMA pieces per operand (row-clamped, chunk-swizzled)
    const char* pa;                 // A tile base + K offset of the tile to stage next (wave-uniform)
    const char* pb;
    unsigned lcur;                  // LDS byte address of this wave's 1 KiB piece slot in the CURRENT tile's buffer (= the one tile t+2 is staged into)
    // implicit-GEMM convolution (conv_w4.hpp): the A pieces of the tile to stage come from srd.base + soff + va[j] where the tap is
    // inside the image for the row (bit `tapbit` of mask[j]), else from beyond the descriptor's range (-> zeros)
    i32x4 srd;
    unsigned soff, tapmask;
    unsigned mask[8];
    // ... with the nearest-2x upsample folded in (CONV = 2): the source of output (2a+e, 2b+f) under tap (dh, dw) is input row a + dH[e], column
    // b + dW[f] (each -1, 0 or +1 by the tap): par[j] = e | f << 1 of the row, uh / uw = the two byte offsets of the current tap
    unsigned par[8];
    unsigned uh[2], uw[2];
};

__device__ __forceinline__ bf16x8_t w4_frag(const Ctx& c, unsigned off) { return *reinterpret_cast<const bf16x8_t*>(c.smem + off); }

// piece q (0..15) of a K tile inside its buffer: q < 8 -> A piece q, else W piece q - 8. Piece j of a wave = rows 32 j + 8 w + [0,8).
#define W4_PIECE_LOFF(q) (((q) >> 3) * OPER_BYTES + ((q) & 7) * 4096)

// One K tile = 128 MFMAs g = 0..127: k-step g >> 6, accumulator tile n = g & 63 = (i, j) = (n >> 3, n & 7). The body exists twice per
// operand order: staging tile t+2 (tiles 0 .. nk-3) and not staging (the last two); the buffers alternate through run-time offsets and
// the last tile reads fragments nobody uses. (Five prologue / tail variants inlined behind one another made hipcc's register allocation
// spill — through a[0:3], this file's accumulator; re-staging the last K tile instead of a second body cost +15 % fabric reads: the
// copies always miss the L2.)
// (plain recursive templates, not generic lambdas: clang does not capture a local for an asm operand inside a generic lambda)
#define W4_X(g) (SWAP ? c.fb[(g) >> 6][(g) & 7] : c.fa[(g) >> 6][((g) & 63) >> 3])
#define W4_Y(g) (SWAP ? c.fa[(g) >> 6][((g) & 63) >> 3] : c.fb[(g) >> 6][(g) & 7])
#define W4_PAIR(g) W4_MFMA2((g) & 63, W4_X(g), W4_Y(g), ((g) + 1) & 63, W4_X((g) + 1), W4_Y((g) + 1))
#define W4_PAIR_DMA(g, voff, sbase, loff) \
    W4_MFMA2_DMA((g) & 63, W4_X(g), W4_Y(g), ((g) + 1) & 63, W4_X((g) + 1), W4_Y((g) + 1), voff, sbase, c.lcur, loff)
#define W4_PAIR_DMAB(g, voff, loff) \
    W4_MFMA2_DMAB((g) & 63, W4_X(g), W4_Y(g), ((g) + 1) & 63, W4_X((g) + 1), W4_Y((g) + 1), voff, c.srd, c.soff, c.lcur, loff)

// ---- the gap plan of a K tile ------------------------------------------------------------------------------------------------------
// The MFMAs go two per asm statement ("slot" k = MFMAs 2k, 2k+1): what rides inside a slot (a DMA piece, between its two MFMAs) or behind
// it (a fragment read, a barrier) is the plan below.
// (r3, measured and removed again: running the four waves in four PHASES — statement grouping shifted by one MFMA, fillers by one slot —
// so that their DMA pieces and fragment reads reach the CU's one vector-memory path / LDS 16 clocks apart instead of in the same clock:
// no change, +-1 %. Neither did removing every wait and barrier of the loop (stale data, timing only). The kernel runs at the package
// power limit — 1.53-1.65 GHz in the PMC passes — where parked cycles cost nothing and wall time follows the ENERGY of a tile:
// profiles/r3_gemm_w4.md.)
#ifndef W4_PLAN
#define W4_PLAN 1
#endif
#ifndef W4_ABLATE
#define W4_ABLATE 0      // experiments only: 1 = no DMA inside the loop (stale operands), 2 = no fragment reads, 4 = no barriers, 8 = no vmcnt wait, 16 = no lgkmcnt(0) at barrier A
#endif
#ifndef W4_BAR_B
#define W4_BAR_B 35
#endif
#ifndef W4_BAR_A
#define W4_BAR_A 17
#endif
#ifndef W4_DMA0
#define W4_DMA0 (W4_BAR_A + 2)
#endif
constexpr int P_BAR_A = W4_BAR_A;    // behind this slot: lgkmcnt(0) + barrier (all of buffer t has been read)
constexpr int P_BAR_B = W4_BAR_B;    // behind this slot: counted vmcnt + barrier (tile t+1 landed and visible)
// DMA piece q of tile t+2 rides in plan position dma_pos(q). PLAN 0: one piece in every slot from 19 on (a burst); PLAN 1: the 16 pieces
// spread over positions 19..63 (one per ~3 slots: 1 KiB per ~24 clocks per CU, under the vector-memory path's 64 B/clk; +5 % at 8192^3, +12 %
// at the ffn.2 shape over the burst).
constexpr int dma_pos(int q) { return W4_PLAN == 0 ? W4_DMA0 + q : W4_DMA0 + (q * (63 - W4_DMA0)) / 15; }
constexpr int piece_at(int pos) {
    for (int q = 0; q < 16; ++q)
        if (dma_pos(q) == pos) return q;
    return -1;
}
constexpr int pieces_before_bar_b() {      // pieces of tile t+2 already issued when the wait for tile t+1 comes
    int k = 0;
    for (int q = 0; q < 16; ++q) k += dma_pos(q) <= P_BAR_B ? 1 : 0;
    return k;
}
// fragment reads: F1(t) read r at plan position r (r = 0..15); F0(t+1) read r at position P_F0_FIRST + r
constexpr int P_F0_FIRST = P_BAR_B + 2;
static_assert(P_F0_FIRST + 16 <= 64 && dma_pos(15) <= 63 && 15 < P_BAR_A && dma_pos(0) > P_BAR_A, "the plan must fit the 64 slots");

template <int CONV>
__device__ __forceinline__ unsigned w4_conv_voff(const Ctx& c, int q) {
    unsigned v = c.va[q];
    if constexpr (CONV == 2) v += ((c.par[q] & 1u) ? c.uh[1] : c.uh[0]) + ((c.par[q] & 2u) ? c.uw[1] : c.uw[0]);
    return (c.mask[q] & c.tapmask) ? v : 0xffffffffu;
}

// CONV: 0 = dense GEMM (A by pointer), 1 = implicit-GEMM convolution (A by frame descriptor, conv_w4.hpp), 2 = ... with a folded 2x upsample
template <int g, bool DMA, bool SWAP, int CONV>
__device__ __forceinline__ void w4_gaps(Ctx& c) {
    if constexpr (g < 128) {
        {
            constexpr int k = g >> 1, pos = k;                 // slot = plan position
            constexpr int q = (DMA && !(W4_ABLATE & 1)) ? piece_at(pos) : -1;
            if constexpr (q >= 0 && q < 8) {
                if constexpr (CONV != 0) W4_PAIR_DMAB(g, w4_conv_voff<CONV>(c, q & 7), W4_PIECE_LOFF(q & 15));
                else W4_PAIR_DMA(g, c.va[q & 7], c.pa, W4_PIECE_LOFF(q & 15));
            } else if constexpr (q >= 8) W4_PAIR_DMA(g, c.vb[q & 7], c.pb, W4_PIECE_LOFF(q & 15));
            else W4_PAIR(g);
            if constexpr (pos >= 0 && pos < 16 && !(W4_ABLATE & 2)) {     // F1(t): the W fragments first (segment 2's first 8 MFMAs need all eight)
                if constexpr (pos < 8) c.fb[1][pos & 7] = w4_frag(c, c.rb1 + (pos & 7) * 2048);
                else c.fa[1][pos & 7] = w4_frag(c, c.ra1 + (pos & 7) * 2048);
            }
            if constexpr (k == P_BAR_A) {
                if constexpr ((W4_ABLATE & 20) == 20) asm volatile("" ::: "memory");
                else if constexpr (W4_ABLATE & 16) asm volatile("s_barrier" ::: "memory");
                else if constexpr (W4_ABLATE & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if constexpr (k == P_BAR_B) {
                // tile t+1 complete (this wave's pieces; the pieces of tile t+2 issued so far may stay in flight), then visible to all waves
                constexpr int keep = (DMA && !(W4_ABLATE & 1)) ? pieces_before_bar_b() : 0;
                if constexpr ((W4_ABLATE & 12) == 12) asm volatile("" ::: "memory");
                else if constexpr (W4_ABLATE & 8) asm volatile("s_barrier" ::: "memory");
                else if constexpr (W4_ABLATE & 4) asm volatile("s_waitcnt vmcnt(%c0)" ::"n"(keep) : "memory");
                else asm volatile("s_waitcnt vmcnt(%c0)\n\ts_barrier" ::"n"(keep) : "memory");
            }
            if constexpr (pos >= P_F0_FIRST && pos < P_F0_FIRST + 16 && !(W4_ABLATE & 2)) {
                constexpr int r = pos - P_F0_FIRST;
                if constexpr (r < 8) c.fb[0][r & 7] = w4_frag(c, c.rb0n + (r & 7) * 2048);
                else c.fa[0][r & 7] = w4_frag(c, c.ra0n + (r & 7) * 2048);
            }
            w4_gaps<g + 2, DMA, SWAP, CONV>(c);
        }
    }
}
// the 16 pieces of one K tile, stand-alone (prologue)
template <int q, int CONV>
__device__ __forceinline__ void w4_stage_all(Ctx& c) {
    if constexpr (q < 16) {
        if constexpr (q < 8) {
            if constexpr (CONV != 0) W4_DMAB(w4_conv_voff<CONV>(c, q & 7), c.srd, c.soff, c.lcur, W4_PIECE_LOFF(q));
            else W4_DMA(c.va[q & 7], c.pa, c.lcur, W4_PIECE_LOFF(q));
        } else W4_DMA(c.vb[q & 7], c.pb, c.lcur, W4_PIECE_LOFF(q));
        w4_stage_all<q + 1, CONV>(c);
    }
}

// ---- epilogue ---------------------------------------------------------------------------------------------------------------------
// SWAP tiles: a lane holds C[m][n .. n+3], m = m0 + wr*128 + 16 i + (lane & 15), n = n0 + wc*128 + 16 j + 4 (lane >> 4).
// !SWAP tiles (K-major V^T): C[m .. m+3][n], m = m0 + wr*128 + 16 i + 4 (lane >> 4), n = n0 + wc*128 + 16 j + (lane & 15).
template <int EPI>
__device__ __forceinline__ f32x4 w4_act(f32x4 v) {
    if (EPI == YUME_EPI_BF16_GELU) {
        const f32x2_t lo = gelu_tanh2(f32x2_t{v[0], v[1]}), hi = gelu_tanh2(f32x2_t{v[2], v[3]});
        v = f32x4{lo[0], lo[1], hi[0], hi[1]};
    }
    if (EPI == YUME_EPI_BF16_GELU_ERF) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = 0.5f * v[q] * (1.0f + erff(v[q] * 0.7071067811865476f));
    }
    return v;
}

// bf16 tile image in LDS: [256 rows][512 B], 16-byte chunk c of row r at chunk c ^ (r & 31); stored as whole rows (1 KiB per wave store).
// rows = output rows of `dst` (tokens m for row-major outputs, features n for the K-major V^T), cols the contiguous index.
// A wave stores rows 64 wave .. +63, two per instruction (lanes 0-31 / 32-63): 8 image reads in flight, then their 8 stores; the per-row
// work is one XOR with a literal (the swizzle) and a 64-bit add (two rows down).
__device__ __forceinline__ void w4_store_image(char* smem, unsigned short* dst, int64_t ld, int row0, int rows_valid, int col0, int cols_valid) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, rh = lane >> 5;
    if (cols_valid == 256) {                                       // (workgroup-uniform) whole rows: every tile but the N / M-edge ones of a ragged matrix
        // row r = 64 wave + 2 it + rh: r & 31 = 2 (it & 15) + rh, so the chunk is (l31 ^ rh) ^ 2 (it & 15)
        const unsigned lrow = (unsigned)((wave * 64 + rh) * 512), lch = (unsigned)((l31 ^ rh) << 4);
        char* o = reinterpret_cast<char*>(dst + col0 + l31 * 8) + (int64_t)(row0 + wave * 64 + rh) * ld * 2;
        const int64_t step = ld * 4;                               // two rows down
        const int rleft = rows_valid - wave * 64 - rh;             // this lane's row 2 it + .. is inside the matrix iff 2 it < rleft
        auto rows = [&](auto pred) {
            constexpr bool PRED = decltype(pred)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x4 d[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int it = g * 8 + u;
                    d[u] = *reinterpret_cast<const u32x4*>(smem + lrow + it * 1024 + (lch ^ (unsigned)(32 * (it & 15))));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int it = g * 8 + u;
                    if (!PRED || 2 * it < rleft) *reinterpret_cast<u32x4*>(o + it * step) = d[u];
                }
            }
        };
        if (rows_valid == 256) rows(std::false_type{});            // (workgroup-uniform)
        else rows(std::true_type{});
        return;
    }
    unsigned short* out = dst + col0 + l31 * 8;
    const int col_left = cols_valid - l31 * 8;                     // elements of this lane's chunk inside the matrix (ragged last chunk)
#pragma unroll 2
    for (int it = 0; it < 32; ++it) {
        const int r = wave * 64 + it * 2 + rh;
        const u32x4 d = *reinterpret_cast<const u32x4*>(smem + r * 512 + (((l31 ^ r) & 31) << 4));
        if (r < rows_valid && col_left > 0) {
            unsigned short* o = out + (int64_t)(row0 + r) * ld;
            if (col_left >= 8) {
                *reinterpret_cast<u32x4*>(o) = d;
            } else {
#pragma unroll
                for (int q = 0; q < 7; ++q)
                    if (q < col_left) o[q] = (unsigned short)(d[q >> 1] >> (16 * (q & 1)));
            }
        }
    }
}

// Image addresses of a lane's 64 accumulator tiles without per-tile arithmetic. A tile (a, b) — a = its 16-row block of the image, b = its
// 16-column block — puts the lane's 4 consecutive columns at row R = 128 rsel + 16 a + l15, chunk (16 csel + 2 b + (l4 >> 1)) ^ (R & 31),
// half l4 & 1. With R & 31 = 16 (a & 1) + l15 the XOR splits into 16 (csel ^ (a & 1)) and (2 b) ^ ((l4 >> 1) ^ l15): one VGPR per b for even
// a, one for odd a, and a * 8192 in the instruction's offset field.
struct ImageOff { unsigned e[8], o[8]; };
__device__ __forceinline__ ImageOff w4_image_offsets(int rsel, int csel, int l15, int l4) {
    ImageOff r;
    const unsigned base = (unsigned)((rsel * 128 + l15) * 512 + (l4 & 1) * 8), ux = (unsigned)(((l4 >> 1) ^ l15) << 4);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const unsigned ob = (unsigned)(32 * b) ^ ux;
        r.e[b] = base + 256u * (unsigned)csel + ob;
        r.o[b] = base + 256u * (unsigned)(csel ^ 1) + ob;
    }
    return r;
}

template <int EPI, bool SWAP>
__device__ __forceinline__ void w4_epilogue(const Problem& p, const Epilogue& e, int m0, int n0, char* smem) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    constexpr bool BF16_OUT = EPI == YUME_EPI_BF16 || EPI == YUME_EPI_BF16_GELU || EPI == YUME_EPI_BF16_GELU_ERF || EPI == YUME_EPI_BF16_SPLITT ||
                              EPI == EPI_BF16_ADD || EPI == EPI_BF16_TSPLIT;
    if constexpr (!SWAP) {
        // K-major V^T tile: image rows = features n, columns = tokens m; whole rows of up to 256 tokens leave as 16-byte stores
        float bn[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bn[j] = e.bias ? e.bias[min(n0 + wc * 128 + 16 * j + l15, p.N - 1)] : 0.f;
        const ImageOff io = w4_image_offsets(wc, wr, l15, l4);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        static_for<0, 64>([&](auto tt) {
            constexpr int T = decltype(tt)::value, i = T >> 3, j = T & 7;
            const f32x4 v = acc_tile<T>();
            u32x2 o;
            o[0] = pack_bf16x2(v[0] + bn[j], v[1] + bn[j]);
            o[1] = pack_bf16x2(v[2] + bn[j], v[3] + bn[j]);
            *reinterpret_cast<u32x2*>(smem + ((j & 1) ? io.o[i] : io.e[i]) + j * 8192) = o;
        });
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        w4_store_image(smem, e.outT + (int64_t)(n0 - e.n_split) * e.ldt, e.ldt, 0, min(256, p.N - n0), m0, min(256, p.M - m0));
        return;
    } else {
        if (BF16_OUT && !p.epi_direct && (e.ldo % 8) == 0) {         // (workgroup-uniform)
            f32x4 b[8];
            if (n0 + 256 <= p.N) {                                   // (workgroup-uniform)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    b[j] = e.bias ? *reinterpret_cast<const f32x4*>(e.bias + n0 + wc * 128 + 16 * j + 4 * l4) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = n0 + wc * 128 + 16 * j + 4 * l4;
                    b[j] = (e.bias && n + 3 < p.N) ? *reinterpret_cast<const f32x4*>(e.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            const ImageOff io = w4_image_offsets(wr, wc, l15, l4);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TRACE_STAMP(3);
            static_for<0, 8>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                u32x2 a2[8];
                if constexpr (EPI == EPI_BF16_ADD) {              // the shortcut the conv output is added to: 8 loads in flight per row block
                    const int r = wr * 128 + 16 * i + l15;
                    const int nl = n0 + wc * 128 + 4 * l4;
                    const unsigned short* ap = e.add + (int64_t)min(m0 + r, p.M - 1) * e.ldadd + nl;
#pragma unroll
                    for (int j = 0; j < 8; ++j)          // (columns beyond N — an N edge tile — are never stored: nothing is read for them)
                        a2[j] = nl + 16 * j + 3 < p.N ? *reinterpret_cast<const u32x2*>(ap + 16 * j) : u32x2{0u, 0u};
                }
                static_for<0, 8>([&](auto jj) {
                    constexpr int j = decltype(jj)::value;
                    f32x4 v = w4_act<EPI>(acc_tile<i * 8 + j>() + b[j]);
                    if constexpr (EPI == EPI_BF16_ADD) {
                        v[0] += bf16_to_f32((unsigned short)(a2[j][0] & 0xffffu));
                        v[1] += bf16_to_f32((unsigned short)(a2[j][0] >> 16));
                        v[2] += bf16_to_f32((unsigned short)(a2[j][1] & 0xffffu));
                        v[3] += bf16_to_f32((unsigned short)(a2[j][1] >> 16));
                    }
                    u32x2 o;
                    o[0] = pack_bf16x2(v[0], v[1]);
                    o[1] = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(smem + ((i & 1) ? io.o[j] : io.e[j]) + i * 8192) = o;
                });
            });
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TRACE_STAMP(4);
            if constexpr (EPI == EPI_BF16_TSPLIT) {
                // out row m=(t,hw), col n=(j,c) -> out[((2t+j)*HW + hw), c]: a tile lies in ONE frame t and one channel half j, so its rows stay
                // consecutive. The contract (conv_w4_kernel): a tile never spans two frames — a frame takes ceil(HW / 256) tiles, m0 is the
                // tile's first row inside the whole matrix and p.M the END OF ITS FRAME ((t+1)*HW), so min(256, p.M - m0) also trims the ragged
                // last tile of a frame whose HW is not a multiple of 256; (N/2) % 256 == 0 is checked by the caller
                const int ch = p.N >> 1, jh = n0 >= ch ? 1 : 0, t = m0 / e.hw;
                w4_store_image(smem, reinterpret_cast<unsigned short*>(e.out), e.ldo, m0 + (t + jh) * e.hw, min(256, p.M - m0), n0 - jh * ch, 256);
            } else {
                w4_store_image(smem, reinterpret_cast<unsigned short*>(e.out), e.ldo, m0, min(256, p.M - m0), n0, min(256, p.N - n0));
            }
            return;
        }
        // whole columns (workgroup-uniform); rows beyond M — the last row of tiles of a ragged M — read row M - 1 and store nothing (that tile
        // is among the last to finish: on the generic guarded path below its epilogue took 64 - 76 us instead of 20 and set the end of the launch)
        if (EPI == YUME_EPI_RESID && n0 + 256 <= p.N) {
            // fp32 residual stream, in place: x += (acc + bias) * gate. The loads of row block i + 2 (8 x vectors, 8 gate vectors of the wave's
            // 128 columns) are issued behind the stores of block i: two blocks (32 KiB per wave) are in flight while one is combined.
            f32x4 b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                b[j] = e.bias ? *reinterpret_cast<const f32x4*>(e.bias + n0 + wc * 128 + 16 * j + 4 * l4) : f32x4{0.f, 0.f, 0.f, 0.f};
            const int mrow = m0 + wr * 128 + l15, ncol = n0 + wc * 128 + 4 * l4;
            const bool ragged_m = m0 + 256 > p.M;                      // (workgroup-uniform)
            const int mlast = p.M - 1 - mrow;                          // row block i is inside the matrix iff 16 i <= mlast
            float* const xo = reinterpret_cast<float*>(e.out) + (int64_t)mrow * e.ldo + ncol;
            const int64_t xstep = 16 * e.ldo;                          // floats between row blocks
            // row block i of this lane: its offset from xo in floats (clamped to the matrix's last row in a ragged tile)
            auto xoff = [&](int i) -> int64_t { return ragged_m ? (int64_t)(min(16 * i, mlast) ) * e.ldo : i * xstep; };
            const bool gated = e.gate != nullptr;                      // (workgroup-uniform)
            bool rowed = gated && e.row_idx != nullptr;                // (wave-uniform)
            int ridx[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ridx[i] = rowed ? e.row_idx[min(mrow + 16 * i, p.M - 1)] : 0;
            f32x4 x[2][8], g[2][8];
            auto issue_x = [&](auto ii, auto bb) {
                constexpr int i = decltype(ii)::value, bf = decltype(bb)::value;
#pragma unroll
                for (int j = 0; j < 8; ++j) x[bf][j] = *reinterpret_cast<const f32x4*>(xo + xoff(i) + 16 * j);
            };
            auto issue_g = [&](auto ii, auto bb) {
                constexpr int i = decltype(ii)::value, bf = decltype(bb)::value;
                if (rowed || (gated && i < 2)) {                       // one gate row for every token: loaded once per buffer
                    const float* gp = e.gate + (int64_t)ridx[i] * e.gate_stride + ncol;
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[bf][j] = *reinterpret_cast<const f32x4*>(gp + 16 * j);
                }
            };
            // the x loads of the first two row blocks go out before anything waits for the row indices
            issue_x(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            issue_x(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            if (rowed) {
                // the gate rows follow the tokens' timestep segments: the 128 rows of a wave almost always share one, and then its 8 vectors
                // are loaded once (as for an un-indexed gate) instead of once per row block — half of the epilogue's load instructions
                const int r0 = __builtin_amdgcn_readfirstlane(ridx[0]);
                bool same = true;
#pragma unroll
                for (int i = 0; i < 8; ++i) same &= ridx[i] == r0;
                if (__all(same)) rowed = false;
            }
            issue_g(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            issue_g(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            TRACE_STAMP(3);                                            // (trace builds) the loads of the first two row blocks are issued
            static_for<0, 8>([&](auto ii) {
                constexpr int i = decltype(ii)::value, bf = i & 1;
                static_for<0, 8>([&](auto jj) {
                    constexpr int j = decltype(jj)::value;
                    const f32x4 v = acc_tile<i * 8 + j>() + b[j];
                    if (gated) x[bf][j] += v * g[bf][j];
                    else x[bf][j] += v;
                    if (!ragged_m || 16 * i <= mlast) *reinterpret_cast<f32x4*>(xo + i * xstep + 16 * j) = x[bf][j];
                });
                if constexpr (i + 2 < 8) {
                    issue_x(std::integral_constant<int, i + 2>{}, std::integral_constant<int, bf>{});
                    issue_g(std::integral_constant<int, i + 2>{}, std::integral_constant<int, bf>{});
                }
                if (i == 0) TRACE_STAMP(4);                            // first row block: loads landed, combined, its stores issued
            });
            TRACE_STAMP(5);                                            // all eight row blocks combined, every store issued
#ifdef YUME_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (trace builds only) ... and drained
            TRACE_STAMP(6);
#endif
            return;
        }
        static_for<0, 64>([&](auto tt) {
            constexpr int T = decltype(tt)::value, i = T >> 3, j = T & 7;
            const f32x4 v = acc_tile<T>();
            const int m = m0 + wr * 128 + 16 * i + l15, n = n0 + wc * 128 + 16 * j + 4 * l4;
            if (m < p.M && n < p.N) store_row4<EPI>(v, m, n, p, e);
        });
    }
}

// the K loop: tiles 0 .. nk-3 stage the tile two ahead, the last two stage nothing (a second instance of the body)
template <bool DMA, bool SWAP>
__device__ __forceinline__ void w4_loop(Ctx& c, int t0, int t1, unsigned lbase) {
    for (int t = t0; t < t1; ++t) {
        w4_gaps<0, DMA, SWAP, 0>(c);
        if constexpr (DMA) {
            c.pa += 128;
            c.pb += 128;
        }
        c.lcur = lbase + (((t + 1) & 1) ? BUF_BYTES : 0);
        c.ra1 ^= BUF_BYTES;
        c.rb1 ^= BUF_BYTES;
        c.ra0n ^= BUF_BYTES;
        c.rb0n ^= BUF_BYTES;
    }
}

// K tiles [kt0, kt0 + nk) of the tile at (m0, n0) (nk >= 3): the whole K range for a data-parallel tile, a slice for a stream-K piece
template <bool SWAP>
__device__ __forceinline__ void w4_mainloop(const Problem& p, const PlainA& al, char* smem, int m0, int n0, int kt0, int nk) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    Ctx c;
    c.smem = smem;
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;
    c.lcur = lbase;
    c.pa = reinterpret_cast<const char*>(al.A + (int64_t)m0 * al.lda) + (int64_t)kt0 * 128;
    c.pb = reinterpret_cast<const char*>(p.W + (int64_t)n0 * p.ldw) + (int64_t)kt0 * 128;
    const unsigned lda_b = (unsigned)al.lda * 2u, ldw_b = (unsigned)p.ldw * 2u;
    // DMA piece j of this wave: rows 32 j + 8 wave + (lane >> 3) of the tile, 16-byte chunk (lane & 7) ^ (row & 7) of the 128-byte K slice
    {
        const int rl = 8 * wave + (lane >> 3);
        const unsigned ch = (unsigned)(((lane & 7) ^ (rl & 7)) << 4);
        const int mleft = p.M - 1 - m0, nleft = p.N - 1 - n0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 32 * j + rl;
            c.va[j] = (unsigned)min(r, mleft) * lda_b + ch;
            c.vb[j] = (unsigned)min(r, nleft) * ldw_b + ch;
        }
    }
    // fragment reads: row (lane & 15) of fragment i (rows 16 i + ...) of this wave's half, logical chunk 4 ks + (lane >> 4), swizzled
    const unsigned sw0 = (unsigned)(((lane >> 4) ^ (lane & 7)) << 4), sw1 = (unsigned)(((4 | (lane >> 4)) ^ (lane & 7)) << 4);
    const unsigned rowa = wr * 16384 + (lane & 15) * 128, rowb = OPER_BYTES + wc * 16384 + (lane & 15) * 128;
    // ---- prologue: tiles 0 and 1 in flight, accumulators zeroed under their latency, F0(0) read ----
    w4_stage_all<0, 0>(c);
    c.pa += 128;
    c.pb += 128;
    c.lcur = lbase + BUF_BYTES;
    w4_stage_all<0, 0>(c);
    c.pa += 128;
    c.pb += 128;
    c.lcur = lbase;
    acc_zero<0, 256>();
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        c.fb[0][r] = w4_frag(c, rowb + sw0 + r * 2048);
        c.fa[0][r] = w4_frag(c, rowa + sw0 + r * 2048);
    }
    c.ra1 = rowa + sw1;
    c.rb1 = rowb + sw1;
    c.ra0n = BUF_BYTES + rowa + sw0;
    c.rb0n = BUF_BYTES + rowb + sw0;
    // ---- main loop ----
    w4_loop<true, SWAP>(c, 0, nk - 2, lbase);
    w4_loop<false, SWAP>(c, nk - 2, nk, lbase);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave is done with its fragments: the epilogue reuses the LDS
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // last MFMA results -> v_accvgpr_read
}

// ---- split-K tail (r6) ----------------------------------------------------------------------------------------------------------------
// The block's N = 3072 GEMMs are 444 tiles = 1.73 rounds of the 256 CUs, and the row split of the N = 9216 / 14336 ones leaves a remainder
// launch: the last, partly filled round costs a whole round (profiles/r6_trace_gemm_epilogue_report.txt: o 16 %, ffn.2 14 % idle tail per CU).
// Hybrid schedule: the first sk_dp tiles of the order (whole rounds) stay one tile per workgroup; each of the LAST R = sk_tiles tiles is cut
// along K so that one more full round of workgroups finishes them together:
//   * R > CUs / 2 (the N = 3072 shapes, R = 188): two slices. R "head" workgroups take K tiles [0, L_h) of one tile each; the remaining CUs
//     take the tails [L_h, nk), q = ceil(R / (CUs - R)) of them one after the other, with L_t = nk / (q + 1): heads and tail workgroups
//     finish together after q / (q + 1) of a tile time (R = 188: q = 3, 0.75 instead of 1);
//   * R <= CUs / 2 (QKV: 52, ffn.0: 24): s = CUs / R equal slices of every tile, one piece per workgroup, all at once (1 / s of a tile time).
//   (first build of the round, measured and replaced: CONTIGUOUS iteration ranges — classic stream-K, one partial per workgroup. Correct, and
//   8 % SLOWER than whole tiles on ffn.2: neighbours sit at different K offsets, nothing is shared in the XCD's L2, and the tail streams
//   its operands at HBM rate, 2.8 us per K tile instead of 1.4 — profiles/r6_gemm_streamk_contiguous_first_run.log. Here every workgroup of a
//   group walks the SAME K range in step with its neighbours, as the whole-tile rounds do.)
//   * a tail slice PUBLISHES its 256 x 256 fp32 partial (write-through `sc1` stores into the caller's scratch, every wave drains, one lane
//     raises the slice's flag); the head FINISHES the tile: it polls the flags of the tile's tail slices, one agent-scope acquire each, adds
//     the partials IN SLICE ORDER into its accumulators (a fixed summation order whatever the timing: results are run-to-run identical) and
//     runs the tile's epilogue. Flags return to zero behind their reader (a launch leaves the scratch as it found it: no memset, capturable).
// Every spin is bounded (a timeout raises the scratch's error word instead of hanging). Placement-independent: every workgroup of the tail
// is resident (one per CU) and none waits on a workgroup that waits (MI355X_MICROARCH.md "Workgroup dispatch ..."; cdna_hip_programming.md
// Guideline 16, recipe R1).
constexpr int SK_SLOT_BYTES = 64 * NTHR_W4 * 16;       // one partial tile: 64 accumulator tiles x 256 lanes x 16 bytes = 256 KiB
constexpr int SK_FLAG_STRIDE = 64;                     // one flag word per 64-byte line
constexpr int SK_MIN_KT = 4;                           // a slice is at least this many K tiles (the loop needs three)
constexpr int SK_MAX_SLOTS = 256;
constexpr unsigned SK_SPIN_LIMIT = 1u << 22;           // polls of ~1 us each before a finisher gives up

__host__ __device__ __forceinline__ int64_t sk_workspace_bytes(int slots) { return (int64_t)slots * (SK_SLOT_BYTES + SK_FLAG_STRIDE) + 64; }

// (the slot is addressed as a wave-uniform base + a 32-bit per-lane offset that is laundered once per piece: with 64-bit per-lane pointers the
// compiler hoisted all 64 tile addresses of a slot out of the piece loop and spilled them across the K loop)
template <int T0, int N>
__device__ __forceinline__ void sk_publish_tiles(const char* slot, unsigned lane_off) {
    if constexpr (N > 0) {
        const f32x4 v = acc_tile<T0>();
        // write-through: the payload is visible at the agent's coherence point once this wave's vmcnt drains (Guideline 16 R1)
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(lane_off + (unsigned)(T0 * (NTHR_W4 * 16))), "v"(v), "s"(slot) : "memory");
        sk_publish_tiles<T0 + 1, N - 1>(slot, lane_off);
    }
}
template <int T0, int N>
__device__ __forceinline__ void sk_gather_tiles(const char* slot, unsigned lane_off) {
    if constexpr (N > 0) {
        f32x4 d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = *reinterpret_cast<const f32x4*>(slot + (lane_off + (unsigned)((T0 + u) * (NTHR_W4 * 16))));
        static_for<0, 8>([&](auto uu) {
            constexpr int u = decltype(uu)::value;
            acc_tile_add<T0 + u>(d[u]);
        });
        sk_gather_tiles<T0 + 8, N - 8>(slot, lane_off);
    }
}

// what follows a piece's K loop: a tail slice (slot >= 0) publishes; a head (nslice > 1) gathers the tile's tail slices; then the epilogue
template <bool SWAP>
__device__ __forceinline__ void w4_finish_piece(const Problem& p, const Epilogue& e, int epi, char* smem, int m0, int n0, int slot, int t, int nslice) {
    typedef unsigned gu32;                                            // (the flag words: accessed with agent-scope atomics only)
    char* const slots = p.sk_ws;
    gu32* const flags = reinterpret_cast<gu32*>(p.sk_ws + (int64_t)SK_MAX_SLOTS * SK_SLOT_BYTES);
    unsigned lane_off = threadIdx.x * 16u;
    asm volatile("" : "+v"(lane_off));                                 // (not a loop invariant: see sk_publish_tiles)
    if (slot >= 0) {
        sk_publish_tiles<0, 64>(slots + (int64_t)slot * SK_SLOT_BYTES, lane_off);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // EVERY storing wave drains
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flags + slot * (SK_FLAG_STRIDE / 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (nslice > 1) {
        for (int j = 1; j < nslice; ++j) {                               // the tile's tail slices, in order
            const int q = (j - 1) * p.sk_tiles + t;
            if (threadIdx.x == 0) {
                gu32* f = flags + q * (SK_FLAG_STRIDE / 4);
                unsigned spins = 0;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > SK_SPIN_LIMIT) {                         // never hang the device: flag the scratch, go on with what is there
                        __hip_atomic_store(flags + SK_MAX_SLOTS * (SK_FLAG_STRIDE / 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            sk_gather_tiles<0, 64>(slots + (int64_t)q * SK_SLOT_BYTES, lane_off);
            __syncthreads();                                              // every wave has read the slot
            if (threadIdx.x == 0) __hip_atomic_store(flags + q * (SK_FLAG_STRIDE / 4), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_nop 15" ::: "memory");                           // v_accvgpr_write -> v_accvgpr_read of the epilogue
    }
    TRACE_STAMP(1);
    if constexpr (!SWAP) {
        w4_epilogue<YUME_EPI_BF16_SPLITT, false>(p, e, m0, n0, smem);
    } else {
        switch (epi) {
#ifndef W4_EXPERIMENT
            case YUME_EPI_BF16_GELU: w4_epilogue<YUME_EPI_BF16_GELU, true>(p, e, m0, n0, smem); break;
            case YUME_EPI_BF16_GELU_ERF: w4_epilogue<YUME_EPI_BF16_GELU_ERF, true>(p, e, m0, n0, smem); break;
            case YUME_EPI_F32: w4_epilogue<YUME_EPI_F32, true>(p, e, m0, n0, smem); break;
            case YUME_EPI_RESID: w4_epilogue<YUME_EPI_RESID, true>(p, e, m0, n0, smem); break;
            case YUME_EPI_BF16_SPLITT: w4_epilogue<YUME_EPI_BF16_SPLITT, true>(p, e, m0, n0, smem); break;
#endif
            default: w4_epilogue<YUME_EPI_BF16, true>(p, e, m0, n0, smem); break;
        }
    }
}

// ONE kernel for every epilogue (the epilogue id is a kernel argument: the 128-MFMA loop body exists in 2 operand orders x {staging, not
// staging} already, and instances of it that meet at a join made hipcc's register allocation spill through the accumulators). A
// data-parallel workgroup is the one-piece case of the stream-K walk: the K loop exists once per operand order.
template <int UNUSED = 0>      // (a template so that the header can be included by several translation units)
__global__ __launch_bounds__(NTHR_W4, 1) void gemm_w4_kernel(Problem p, PlainA al, Epilogue e, int epi) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_W4];
    TRACE_STAMP(0);
    const int nk = p.K / BK;
    const int ndp = p.sk_wgs ? p.sk_dp : p.tiles_m * p.tiles_n;       // tiles walked whole, one per workgroup
    // this workgroup's pieces: piece i = (tile t, slice j) for i in [pc0, pc1) of a list; a data-parallel workgroup has the one whole tile
    int pc0 = 0, pc1 = 1, R = 1, tile0, head_t = 0;
    bool tails = false;
    if ((int)blockIdx.x < ndp) {
        int start, count;
        xcd_chunk(ndp, blockIdx.x & 7, start, count);
        tile0 = start + (blockIdx.x >> 3);
    } else {
        const int sk = (int)blockIdx.x - ndp, per = p.sk_wgs >> 3;      // band position: XCD (sk & 7) holds positions per * xcd .. + per - 1
        const int w = (sk & 7) * per + (sk >> 3);
        R = p.sk_tiles;
        tile0 = ndp;
        if (w < R) {
            head_t = w;                                                 // slice 0 of tile w: finishes it
        } else {
            tails = true;                                               // tail slices (w - R) q .. + q - 1 of the (s - 1) R, slice-major
            const int ntail = (p.sk_s - 1) * R;
            pc0 = (w - R) * p.sk_q;
            pc1 = min(ntail, pc0 + p.sk_q);
            if (pc0 >= pc1) return;                                     // (a workgroup that only rounds the launch up to whole XCD bands)
        }
    }
    for (int pc = pc0; pc < pc1; ++pc) {
        int t = head_t, j = 0, k0 = 0, k1 = nk, slot = -1, nslice = 1;
        if (p.sk_wgs && (int)blockIdx.x >= ndp) {
            if (tails) {
                j = 1 + pc / R;
                t = pc - (j - 1) * R;
                k0 = p.sk_lh + (j - 1) * p.sk_lt;
                k1 = j == p.sk_s - 1 ? nk : k0 + p.sk_lt;
                slot = pc;
            } else {
                k1 = p.sk_lh;
                nslice = p.sk_s;
            }
        }
        int m0, n0;
        tile_origin(p, tile0 + t, m0, n0);
        if (epi == YUME_EPI_BF16_SPLITT && n0 >= e.n_split) {
            w4_mainloop<false>(p, al, smem, m0, n0, k0, k1 - k0);
            w4_finish_piece<false>(p, e, epi, smem, m0, n0, slot, t, nslice);
        } else {
            w4_mainloop<true>(p, al, smem, m0, n0, k0, k1 - k0);
            w4_finish_piece<true>(p, e, epi, smem, m0, n0, slot, t, nslice);
        }
        if (pc + 1 < pc1) __syncthreads();                              // (LDS is free before the next piece's first DMA lands)
    }
    TRACE_STAMP(2);
}

// shapes the kernel takes: at least THREE K tiles (two are in flight before the loop, the loop body stages a third); 32-bit per-lane source offsets inside a tile's 256 rows
inline bool w4_applies(const Problem& p, int64_t lda, int epi) {
    if (p.K < 3 * BK) return false;
#ifdef W4_EXPERIMENT
    if (epi != YUME_EPI_BF16) return false;
#endif
    if (255ll * lda * 2 + 128 >= (1ll << 32) || 255ll * p.ldw * 2 + 128 >= (1ll << 32)) return false;
    return epi == YUME_EPI_BF16 || epi == YUME_EPI_BF16_GELU || epi == YUME_EPI_BF16_GELU_ERF || epi == YUME_EPI_F32 || epi == YUME_EPI_RESID ||
           epi == YUME_EPI_BF16_SPLITT;
}

// split-K plan of a launch's tail (host), or sk_wgs = 0
inline void w4_sk_plan(Problem& p, void* ws, int64_t ws_bytes) {
    p.sk_dp = 0; p.sk_tiles = 0; p.sk_wgs = 0; p.sk_s = 1; p.sk_q = 1; p.sk_lh = 0; p.sk_lt = 0; p.sk_ws = nullptr;
    static const bool on = [] { const char* v = getenv("YUME_GEMM_SK"); return !v || atoi(v) != 0; }();
    static const int ncu = [] {
        int d = 0, n = 0;
        if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess) n = 0;
        return n;
    }();
    if (!on || ws == nullptr || ncu < 64 || ncu > SK_MAX_SLOTS || (ncu & 7) != 0) return;
    if (ws_bytes < sk_workspace_bytes(SK_MAX_SLOTS) || ((uintptr_t)ws & 15) != 0) return;
    const int T = p.tiles_m * p.tiles_n, nk = p.K / BK;
    const int R = T % ncu;
    if (R == 0 || T < ncu) return;                                     // whole rounds already, or a launch that does not fill the chip
    // Where it pays (profiles/r6_gemm_splitk_tail_ab.log, 5B shapes, back-to-back launches): ffn.2 (K = 14336, R = 188) 0.729 -> 0.701 ms; the
    // K = 3072 shapes LOSE — o 0.182 -> 0.202, cross-q 0.152 -> 0.172 (a tail slice of 12 K tiles pays a K-loop prologue and a 256 KiB
    // publish, the head a 256 KiB gather: more than the quarter tile they save), QKV 0.466 -> 0.474 and ffn.0 0.708 -> 0.730 (4 / 10 slices
    // of 12 / 4 K tiles). So: long K only, and only the two-slice form. YUME_GEMM_SK_MIN_NK overrides the threshold (A/B runs).
    static const int min_nk = [] { const char* v = getenv("YUME_GEMM_SK_MIN_NK"); return v ? atoi(v) : 96; }();
    if (nk < min_nk) return;
    int s, q, lh, lt;
    if (2 * R > ncu) {                                                 // heads + tails, q tails per tail workgroup
        s = 2;
        q = (R + (ncu - R) - 1) / (ncu - R);
        lt = nk / (q + 1);
        lh = nk - lt;
    } else {                                                           // s equal slices, every piece its own workgroup
        s = ncu / R;
        if (s > nk / SK_MIN_KT) s = nk / SK_MIN_KT;
        q = 1;
        lt = nk / s;
        lh = lt;
    }
    if (s < 2 || lt < SK_MIN_KT || lh < SK_MIN_KT || (s - 1) * R > SK_MAX_SLOTS) return;
    const int nwg = R + ((s - 1) * R + q - 1) / q;
    if (nwg > ncu) return;
    p.sk_tiles = R;
    p.sk_dp = T - R;
    p.sk_s = s; p.sk_q = q; p.sk_lh = lh; p.sk_lt = lt;
    p.sk_wgs = (nwg + 7) / 8 * 8;                                      // whole XCD bands (the surplus workgroups return at once)
    p.sk_ws = (char*)ws;
}

inline int launch_w4(int epi, const Problem& p128, const PlainA& al, const Epilogue& e, hipStream_t st, const char* what, void* ws = nullptr,
                     int64_t ws_bytes = 0) {
    Problem p = p128;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    p.group_m = g_group_m;
    { static const int d = [] { const char* v = getenv("YUME_GEMM_EPI_DIRECT"); return v ? atoi(v) : 0; }(); p.epi_direct = d; }
    w4_sk_plan(p, ws, ws_bytes);
    const unsigned grid = p.sk_wgs ? (unsigned)(p.sk_dp + p.sk_wgs) : (unsigned)(p.tiles_m * p.tiles_n);
    hipLaunchKernelGGL(gemm_w4_kernel<0>, dim3(grid), dim3(NTHR_W4), 0, st, p, al, e, epi);
    YUME_CHECK_LAUNCH(what);
    return YUME_OK;
}

}  // namespace gemm_w4
