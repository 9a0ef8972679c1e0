// attn_fwd7.hip — self-attention forward for head_dim 128 on gfx950, ONE wave per SIMD (bf16 in, fp32 accumulate, bf16 out).
//
// Same function and the same transposed formulation as attn_fwd.hip (S^T = K Q^T, O^T = V^T P^T, a softmax row lives in one
// lane and its lane^32 partner), different schedule. The 8-wave kernel there alternates a matrix phase and a softmax phase
// between the two waves of a SIMD; measured, the partner's VALU work overlaps the MFMAs only by ~45 % (tools/ubench/coissue.hip),
// while VALU work issued BETWEEN the MFMAs of the same wave is almost free. So here:
//   * a workgroup is 4 waves (one per SIMD, the whole 512-entry register file each) and 256 queries; a wave owns TWO blocks
//     of 32 queries, A and B;
//   * a key tile (64 keys) is processed in two phases of 32 MFMAs:
//         phase 1(t):  S_A(t+1) = K(t+1) Q_A^T,  O_A += V^T(t) P_A(t)      with softmax_B(t)   issued between the MFMAs
//         phase 2(t):  S_B(t+1) = K(t+1) Q_B^T,  O_B += V^T(t) P_B(t)      with softmax_A(t+1) issued between the MFMAs
//     the MFMAs of a phase never depend on the VALU work beside them, so dependent-latency chains of the softmax hide
//     under the matrix pipe; one piece (<= ~6 instructions) of the softmax sits in each MFMA gap, written out by hand
//     (sm_piece<I>) and pinned with sched_barriers;
//   * two bodies. ROBUST (any caller): x = s * c - m as one fma per score, the exponentials taken against the block's running base
//     (deferred rescale, P < 2^13); no row maximum is reduced on that path: a partial row sum above 2^13 (inf on the first tile) is
//     what sends a tile to the redo path, which takes the maximum from S, rescales O and l and rebuilds the tile (first tile; then
//     almost never). BASE-FREE (kernel instantiation <true>, AttnArgs::q_prescaled: the caller's Q already carries softmax scale *
//     log2(e), folded in before Q's one bf16 rounding): the scores leave the matrix pipe as the exponents, p = exp2(s) with no shift
//     and no base at all — floating point is scale-invariant (exp2(s - m) and exp2(s) carry the same relative error, m cancels in
//     O / l), so the per-score fma, the per-tile range vote and its branch do not exist. Its guard is ONE check per workgroup at the
//     end (row sums finite and not tiny, every O element finite, LDS vote over the four waves): a workgroup that fails it reruns
//     its key range on the robust body inside the launch (Q^T is still in its AGPRs), so every input has a defined result;
//   * the LDS-DMA pieces of the steady loop ride INSIDE the score-MFMA statements of gaps 0..7 (`s_add_u32 m0, lbase, literal;
//     v_mfma; global_load_lds`): the MFMA is the wait state the M0 write needs, and the 32 destinations of a four-tile trip cost
//     one SGPR instead of 32; the even PV gap's MFMA statement names the NEXT gap's V^T fragment too, so one compiler wait covers
//     the pair. 5.16 (base-free) / 6.3 (robust) instructions per MFMA gap; tests/test_attn7_isa.py holds the budget;
//   * O (a[0:127]) and Q^T (a[128:191]) live in AGPRs this file OWNS: they are named literally in inline asm (MFMAs,
//     v_accvgpr_read/write) and listed as clobbers of every such statement, so hipcc keeps nothing of its own there
//     (tests/test_attn7_isa.py audits the generated code for that). hipcc's own choice for a 512-register kernel puts
//     every accumulator, the scores included, in AGPRs and copies them out for the softmax; with "a"-constrained C++
//     variables instead it shuffles and spills the 512-bit tuples. The scores S, the packed P and the softmax state stay
//     compiler-managed VGPRs. Hazards the compiler cannot see (MFMA result -> VALU / accvgpr read) are covered by
//     distance (>= 16 MFMAs) in the pipelined path and by explicit s_nop pads in the rare paths;
//   * K / V^T tiles arrive by LDS-DMA into 4 + 4 slots of 16 KiB, three tiles ahead, one counted vmcnt + one workgroup
//     barrier per tile; the 16 K fragments of a tile are read from LDS ONCE into AGPRs a[192:255] (right after their last
//     reader, one per gap) and feed the S MFMAs of both blocks; V^T fragments go through a 4-deep VGPR ring.
// LDS images are those of attn_fwd.hip v2/v4 (K: chunk ^ (row & 15); V^T: chunk ^ ((row >> 1) & 7), swizzled on the DMA source).
// Roofline: MFMA bf16 dense; algorithmic work 4*Lq*Lk*128 flop per head.
#include "attn7_core.hpp"

namespace {

// PRE: Q already carries softmax scale * log2(e) (AttnArgs::q_prescaled; scale_log2 is 1): the FAST pieces run first.
template <bool PRE>
__global__ __launch_bounds__(256, 1) void attn_fwd_kernel_v7(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[LDS7 + 16];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int ql = lane & 31;
    const int nt = (p.Lk + KT - 1) / KT;
    TRACE_STAMP(0);
    // block -> (head, query block[, key range]): XCD x (= blockIdx % 8) owns heads x, x+8, ...; its whole blocks come first, then the
    // key-range pieces of the blocks >= tail_qb (hardware dispatches block ids in order: the pieces fill the last, partial round)
    int h, qb, sp = 0, nsp = 1;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int hx = (p.H + 7 - xcd) >> 3;
        const int nmain = hx * p.tail_qb, ntq = p.nqb - p.tail_qb;
        if (idx < nmain) {
            h = xcd + 8 * (idx / p.tail_qb);
            qb = idx % p.tail_qb;
        } else {
            const int j = idx - nmain;
            if (j >= hx * ntq * p.splits) return;
            const int u = j / p.splits;
            sp = j % p.splits;
            nsp = p.splits;
            h = xcd + 8 * (u / ntq);
            qb = p.tail_qb + u % ntq;
        }
    }
    const int t0 = (int)((int64_t)nt * sp / nsp), t1 = (int)((int64_t)nt * (sp + 1) / nsp);
    const int q0 = p.q_lo + qb * QB7 + wave * 64;

    Ctx cx;
    cx.smem = smem;
    cx.lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    cx.lbase = cx.lds0 + wave * 1024;
    cx.c = p.scale_log2;
    cx.keyh = hi << A7_KEYH_SHIFT;
    cx.Lk = p.Lk;
    cx.wave = wave;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) cx.koff[ks] = ql * 256 + (((2 * ks + hi) ^ (ql & 15)) << 4);
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) cx.voff[sg] = VB + ql * 128 + (((2 * sg + hi) ^ ((ql >> 1) & 7)) << 4);

    Dma7 dp;
    dma7_init(dp, p, h, tid);

    Blk A, B;
    if constexpr (PRE) {
        run_keys<true, true>(p, cx, dp, A, B, q0, ql, h, hi, t0, t1, nt, tid);
        // one vote per workgroup (the tile loop is a workgroup affair: shared K / V^T slots, barriers): rerun on the robust pieces?
        const bool ok = block_in_range<OA>(A) & block_in_range<OB>(B);
        int* votes = reinterpret_cast<int*>(smem + LDS7);
        const int wave_ok = __all(ok) ? 1 : 0;        // (outside the lane-0 branch: the vote is over the whole wave)
        if (lane == 0) votes[wave] = wave_ok;
        __syncthreads();
        const int all_ok = votes[0] & votes[1] & votes[2] & votes[3];
        __syncthreads();
        if (__builtin_expect(!__builtin_amdgcn_readfirstlane(all_ok), 0)) {
            cx.c = 1.0f;
            run_keys<false, false>(p, cx, dp, A, B, q0, ql, h, hi, t0, t1, nt, tid);
        }
    } else {
        run_keys<false, true>(p, cx, dp, A, B, q0, ql, h, hi, t0, t1, nt, tid);
    }

    TRACE_STAMP(1);
    if (nsp > 1) {
        const int row0 = p.q_lo + p.tail_qb * QB7;
        const int64_t rows = p.Lq - row0;
        store_partial<OA>(p, A, q0 + ql, h, hi, sp, rows, row0);
        store_partial<OB>(p, B, q0 + 32 + ql, h, hi, sp, rows, row0);
    } else {
        store_block<OA>(p, A, q0 + ql, h, hi);
        store_block<OB>(p, B, q0 + 32 + ql, h, hi);
    }
    TRACE_STAMP(2);
}

}  // namespace

void yume_attn7_launch(const AttnArgs& a, hipStream_t st) {
    // every XCD slot gets ceil(H/8) * (whole blocks + pieces) block ids; surplus ids exit immediately
    const int64_t per = (int64_t)a.tail_qb + (int64_t)(a.nqb - a.tail_qb) * a.splits;
    const dim3 grid((unsigned)(((a.H + 7) / 8) * per * 8));
    if (a.q_prescaled) hipLaunchKernelGGL(attn_fwd_kernel_v7<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attn_fwd_kernel_v7<false>, grid, dim3(256), 0, st, a);
}
