// attn_fwd8.hip — the one-wave-per-SIMD attention of attn_fwd7.hip as PERSISTENT workgroups over ONE CONTINUOUS K / V^T STREAM (r4).
//
// What attn_fwd7 spends outside its steady loop (per 256-query workgroup at the 5B shape, profiles/r3_trace_report.txt: 4.4 us of prologue —
// Q from HBM, the first K tiles —, a first tile at half rate, seven tail tiles on general code at 2.5 instead of 1.5 us, 4.6 us of epilogue,
// 0.9 us of dispatch gap: 8 % of a pass over the keys; for the 512-key cross-attention more than half of it) is the cost of treating every
// (head, query block) as its own launch-let. Here a workgroup is resident for the whole launch (one per CU), draws its items — (head, query
// block[, key range]) — by ticket from its XCD's queue (an XCD owns heads x, x + 8, ...: its 32 CUs sweep the same K / V^T and share them
// in the XCD's L2; an XCD whose queue has run dry steals from the others', so the chip's unequal XCD speeds even out), and NEVER drains its
// pipeline between items:
//   * the K / V^T stream is continuous. The LDS-DMA pieces that the last four tiles of an item used to leave out fetch the FIRST tiles
//     of the next item (of whatever head); the slot of a tile is a global tile counter & 3, so the steady code of attn_fwd7 (compile-time
//     slots, pieces riding in the score MFMAs, one counted wait + one barrier per tile) runs every tile of every item;
//   * the score half of an item's LAST tile already works for the next item: S'(0) = K'(0) Q'^T beside O += V^T(last) P(last). Q' is
//     loaded straight into the AGPRs Q occupied (global_load with an AGPR destination: no VGPRs) in the one gap where they are free —
//     between the last two tiles (bubble 1: one memory latency; the ticket for the item after next is drawn under it);
//   * behind the last tile (bubble 2) the workgroup checks the base-free body's range vote, normalises and stores O^T, zeroes it and
//     goes on: the softmax of the new item's first tile is already half done.
// A key range that ends at the ragged last tile of the sequence is only MASKED (the last two tiles run the masked pieces): the caller
// guarantees (YUME_ATTN_KV_PADDED) that K is readable up to a whole number of 64-key tiles and that V^T's columns up to there hold
// finite values, so a ragged tile is fetched like any other — no clamped sources, no fix-up pass.
// Only the base-free body streams (YUME_ATTN_Q_PRESCALED). A workgroup whose range vote fails reruns that item cold on attn_fwd7's robust
// pieces (run_keys<false, true>) and restarts the stream with the next one: every input has a defined result, as in attn_fwd7.
// Same arithmetic per tile and the same tile order as attn_fwd7: whole query blocks come out bit-identical (tests/test_ops_gpu.py).
// Roofline: MFMA bf16 dense; algorithmic work 4*Lq*Lk*128 flop per head.
#define A7_PIN_MASKED_EXP 1
#include "attn7_core.hpp"

namespace {

struct Item {
    int h, qb, t0, t1, sp, nsp;                 // nsp == 0: no item
};

// items of XCD y's queue, in dispatch order: the whole query blocks head by head, then the key-range pieces of the blocks >= tail_qb
__device__ __forceinline__ int queue_len(const AttnArgs& p, int y) {
    const int hx = (p.H + 7 - y) >> 3;
    return hx * (p.tail_qb + (p.nqb - p.tail_qb) * p.splits);
}
__device__ __forceinline__ Item decode_item(const AttnArgs& p, int ticket, int nt) {
    Item it;
    if (ticket < 0) {
        it.h = it.qb = it.t0 = it.t1 = it.sp = it.nsp = 0;
        return it;
    }
    const int y = ticket >> 24, j = ticket & 0xffffff;
    const int hx = (p.H + 7 - y) >> 3;
    const int nmain = hx * p.tail_qb, ntq = p.nqb - p.tail_qb;
    if (j < nmain) {
        it.h = y + 8 * (j / p.tail_qb);
        it.qb = j % p.tail_qb;
        it.sp = 0;
        it.nsp = 1;
    } else {
        const int u = (j - nmain) / p.splits;
        it.sp = (j - nmain) % p.splits;
        it.nsp = p.splits;
        it.h = y + 8 * (u / ntq);
        it.qb = p.tail_qb + u % ntq;
    }
    it.t0 = (int)((int64_t)nt * it.sp / it.nsp);
    it.t1 = (int)((int64_t)nt * (it.sp + 1) / it.nsp);
    // (integer division runs on the vector ALU: hand the wave-uniform results back to scalar registers explicitly, or hipcc moves every
    // loop-carried scalar that meets them — tile counters, the LDS-DMA source pointers — into VGPRs)
    it.h = __builtin_amdgcn_readfirstlane(it.h);
    it.qb = __builtin_amdgcn_readfirstlane(it.qb);
    it.t0 = __builtin_amdgcn_readfirstlane(it.t0);
    it.t1 = __builtin_amdgcn_readfirstlane(it.t1);
    it.sp = __builtin_amdgcn_readfirstlane(it.sp);
    it.nsp = __builtin_amdgcn_readfirstlane(it.nsp);
    return it;
}
__device__ __forceinline__ const char* uniform_ptr(const char* q) {
    const uint64_t v = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
// thread 0 only: the next ticket (XCD << 24 | index) of the own queue or, when that has run dry, of the others'; -1 when nothing is
// left — a workgroup draws that -1 exactly ONCE (it never draws again), so the last of the `nwg` to draw it knows that nobody will
// touch the counter set any more and writes the zeros back (counters.hpp)
__device__ __forceinline__ int draw_ticket(const AttnArgs& p, int* cnt, int xcd, int nwg) {
    for (int k = 0; k < 8; ++k) {
        const int y = (xcd + k) & 7, n = queue_len(p, y);
        if (n <= 0) continue;
        const int t = atomicAdd(&cnt[y], 1);
        if (t < n) return (y << 24) | t;
    }
    __threadfence();
    if (atomicAdd(&cnt[9], 1) == nwg - 1) {
        for (int k = 0; k < 10; ++k) atomicExch(&cnt[k], 0);
    }
    return -1;
}

// the lane id, rebuilt in two instructions wherever it is needed (volatile: not hoisted, not kept)
__device__ __forceinline__ unsigned fresh_lane() {
    unsigned l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// one LDS-DMA piece of 64 lanes x 4 bytes (glds16's little brother)
__device__ __forceinline__ void glds4(const char* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// Q^T fragments of a block's query rows straight into the AGPRs a[XQ .. XQ+31] (lane (q, hi) holds Q[q][16*ks + 8*hi .. +7] -> a[XQ + 4*ks ..]):
// eight 16-byte loads per lane, SGPR base + per-lane 32-bit offset. Completion is the caller's s_waitcnt vmcnt.
template <int XQ>
__device__ __forceinline__ void load_q_agpr(const char* qbase, unsigned voff) {
    asm volatile(
        "global_load_dwordx4 a[%c2:%c3], %0, %1\n\t"
        "global_load_dwordx4 a[%c4:%c5], %0, %1 offset:32\n\t"
        "global_load_dwordx4 a[%c6:%c7], %0, %1 offset:64\n\t"
        "global_load_dwordx4 a[%c8:%c9], %0, %1 offset:96\n\t"
        "global_load_dwordx4 a[%c10:%c11], %0, %1 offset:128\n\t"
        "global_load_dwordx4 a[%c12:%c13], %0, %1 offset:160\n\t"
        "global_load_dwordx4 a[%c14:%c15], %0, %1 offset:192\n\t"
        "global_load_dwordx4 a[%c16:%c17], %0, %1 offset:224" ::"v"(voff),
        "s"(qbase), "n"(XQ), "n"(XQ + 3), "n"(XQ + 4), "n"(XQ + 7), "n"(XQ + 8), "n"(XQ + 11), "n"(XQ + 12), "n"(XQ + 15), "n"(XQ + 16), "n"(XQ + 19),
        "n"(XQ + 20), "n"(XQ + 23), "n"(XQ + 24), "n"(XQ + 27), "n"(XQ + 28), "n"(XQ + 31)
        : "memory", OWNED_AGPRS);
}

// attn_fwd7's store_block / store_partial read all 64 accumulators of a block into VGPRs first; between two tiles of a stream that costs
// registers the softmax state of BOTH blocks needs (hipcc then spills it, and the reloads wait — vmcnt(0) — behind the next tile's LDS-DMA
// pieces). Same values, same stores, 16 accumulators (one d block) at a time.
template <int XO>
__device__ __forceinline__ void store_block8(const AttnArgs& p, const Blk& x, int q, int h, int hi) {
    const float l_tot = xhalf_sum(x.z.l_run);
    const float inv = 1.0f / l_tot;
    if (q < p.Lq) {
        unsigned short* op = p.O + (int64_t)q * p.ldo + h * D + 4 * hi;
        for_regs<0, 4>([&](auto dbc) {
            constexpr int db = decltype(dbc)::value;
            float o[16];
            for_regs<0, 16>([&](auto r) { o[decltype(r)::value] = agpr_get<XO + 16 * db + decltype(r)::value>() * inv; });
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = o[4 * g + 0], v1 = o[4 * g + 1], v2 = o[4 * g + 2], v3 = o[4 * g + 3];
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
        });
    }
}
template <int XO>
__device__ __forceinline__ void store_partial8(const AttnArgs& p, const Blk& x, int q, int h, int hi, int sp, int64_t rows, int row0) {
    const float l_part = xhalf_sum(x.z.l_run);
    if (q < p.Lq) {
        const int64_t r = q - row0;
        float* po = p.part_o + ((int64_t)sp * rows + r) * ((int64_t)p.H * D) + h * D + 4 * hi;
        for_regs<0, 4>([&](auto dbc) {
            constexpr int db = decltype(dbc)::value;
            float o[16];
            for_regs<0, 16>([&](auto r2) { o[decltype(r2)::value] = agpr_get<XO + 16 * db + decltype(r2)::value>(); });
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(po + 32 * db + 8 * g) = f32x4{o[4 * g + 0], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
        });
        if (hi == 0) {
            float* pm = p.part_ml + (((int64_t)sp * rows + r) * p.H + h) * 2;
            pm[0] = x.z.m_run;
            pm[1] = l_part;
        }
    }
}

// (r4 built a whole-row variant of store_block8 — the O^T block transposed through a per-wave LDS patch into 128-byte row segments, worth
// 1.1 us of the 5-6 us the lane-strided stores cost per item; it made the items that take the robust rerun differ by 1 bf16 ulp between two
// launches on the same data (profiles/r4_attn8_rows_store_bisect.log), the cause was not found, and r5 removed the code instead of parking
// it behind a macro: git show 89876aa:yume_amd/csrc/attn_fwd8.hip has it. tests/test_ops_gpu.py launches the rerun shapes twice and
// compares bits, so a latent race in the shared rerun path would show.)

// One tile of the stream on compile-time slots (TS = global tile counter & 3). KIND 0: attn_fwd7's steady tile. KIND 1: the tile before an
// item's last — its second phase computes softmax_A of the last tile, masked against Lk (jl = first key of that tile). KIND 2: the last
// tile of an item that has a successor: both drains and softmax_B are the last tile's (masked), the score MFMAs and softmax_A are the next
// item's first tile (Q' is in the AGPRs, K'(0) in the cache, the K refill brings K'(1)).
// The key mask of the masked pieces is `keyb + const < Lk` with keyb = first key of the tile + 4 * (lane >> 5): a per-lane value that is
// needed in two tiles per item. Carried across the item it is spilled and comes back through a reload whose wait (vmcnt(0), hipcc cannot
// count the LDS-DMA pieces) lands behind the tile's first pieces. So the masked tiles rebuild it from the lane id in two instructions and
// fold (first key - Lk) into it: the pieces then compare against the constant 0.
__device__ __forceinline__ Ctx masked_ctx(const Ctx& cx, int jl_minus_lk) {
    Ctx cm = cx;
    unsigned l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    cm.keyh = (int)((l >> 5) << A7_KEYH_SHIFT) + jl_minus_lk;
    cm.Lk = 0;
    return cm;
}

template <int TS, int KIND>
__device__ __forceinline__ void tile8(const Ctx& cx, const Dma7& dp, const char*& kg, const char*& vg, int64_t kstep, Blk& A, Blk& B, u32x4 (&ring)[RD], int jl) {
    constexpr int vb = TS * SLOT, nkb = ((TS + 2) & 3) * SLOT;
    constexpr int dk = TS * SLOT, dv = VB + ((TS + 3) & 3) * SLOT;
    if constexpr (KIND != 0 && KIND != 3) {
        // whatever of the softmax state hipcc parked in scratch across the bubble comes back HERE, in front of the counted wait (a reload
        // inside the tile would wait for the tile's own LDS-DMA pieces)
        PIN_BLK(A);
        PIN_BLK(B);
    }
    // KIND 3: the first tile behind an item boundary. Everything it reads was waited for in bubble 1 (vmcnt(0) in front of the boundary
    // tile); in flight are the boundary tile's 8 pieces — and the O^T stores of the item just finished, which a counted wait would have
    // to sit out (one counter for loads and stores). Only the barrier.
    if constexpr (KIND != 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (KIND == 0 || KIND == 3) {
        phase<true, OA, QA, OB, true, true, true, false, true, false, true, -1, dk, dv>(cx, A, B, ring, vb, 0, 0, dp, kg, vg);
        phase<true, OB, QB, OA, true, true, true, false, true, false, false, nkb>(cx, B, A, ring, vb, 0, 0, dp, kg, vg);
    } else if constexpr (KIND == 1) {
        const Ctx cm = masked_ctx(cx, jl - cx.Lk);
        phase<true, OA, QA, OB, true, true, true, false, true, false, true, -1, dk, dv>(cx, A, B, ring, vb, 0, 0, dp, kg, vg);
        phase<true, OB, QB, OA, true, true, true, false, true, true, false, nkb>(cm, B, A, ring, vb, 0, 0, dp, kg, vg);
    } else {
        const Ctx cm = masked_ctx(cx, jl - cx.Lk);
        phase<true, OA, QA, OB, true, true, true, true, true, true, true, -1, dk, dv>(cm, A, B, ring, vb, 0, 0, dp, kg, vg);
        phase<true, OB, QB, OA, true, true, true, true, true, false, false, nkb>(cm, B, A, ring, vb, 0, 0, dp, kg, vg);
    }
    kg += kstep;
    vg += KT * 2;
}
// The four TS instances as a CHAIN of tests, not a switch: the arms of a switch (a binary tree of branches) start with the same pure
// computations on the same values — the exponentials of a phase depend on nothing but the scores — and hipcc's branch folding hoists such a
// common prefix of two sibling arms into their parent: 32 exponentials and their packs live at once, the register file overflows into
// scratch, and the reloads wait (vmcnt(0)) behind the tile's LDS-DMA pieces. In a chain an arm's sibling is the next test.
template <int KIND>
__device__ __forceinline__ void tile8_any(int g, const Ctx& cx, const Dma7& dp, const char*& kg, const char*& vg, int64_t kstep, Blk& A, Blk& B, u32x4 (&ring)[RD], int jl) {
    int ts = __builtin_amdgcn_readfirstlane(g & 3);
    asm volatile("" : "+s"(ts));          // (opaque: the tests below are not folded back into a switch)
    if (ts == 0) tile8<0, KIND>(cx, dp, kg, vg, kstep, A, B, ring, jl);
    asm volatile("" : "+s"(ts));
    if (ts == 1) tile8<1, KIND>(cx, dp, kg, vg, kstep, A, B, ring, jl);
    asm volatile("" : "+s"(ts));
    if (ts == 2) tile8<2, KIND>(cx, dp, kg, vg, kstep, A, B, ring, jl);
    asm volatile("" : "+s"(ts));
    if (ts == 3) tile8<3, KIND>(cx, dp, kg, vg, kstep, A, B, ring, jl);
}

__global__ __launch_bounds__(256, 1) void attn_fwd_kernel_v8(AttnArgs p, int* cnt, int nwg) {
    __shared__ __attribute__((aligned(16))) char smem[LDS7];
    __shared__ int votes[4];                           // the range vote of the four waves
    __shared__ int mail[2];                            // tickets drawn by thread 0, read by everybody behind a barrier
    __shared__ __attribute__((aligned(16))) int junk[4 * 128];   // where the Q' touch (below) drops what it fetched: 2 pieces x 256 B per wave
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int ql = lane & 31;
    const int nt = (p.Lk + KT - 1) / KT;
    const int xcd = blockIdx.x & 7;
    TRACE_STAMP(0);

    Ctx cx;
    cx.smem = smem;
    cx.lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    cx.lbase = cx.lds0 + wave * 1024;
    cx.c = 1.0f;
    cx.keyh = hi << A7_KEYH_SHIFT;
    cx.Lk = p.Lk;
    cx.wave = wave;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) cx.koff[ks] = ql * 256 + (((2 * ks + hi) ^ (ql & 15)) << 4);
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) cx.voff[sg] = VB + ql * 128 + (((2 * sg + hi) ^ ((ql >> 1) & 7)) << 4);

    Dma7 dp;
    dma7_init(dp, p, 0, tid);                          // per-lane piece offsets (the same for every head); kbase / vbase are set per item
    const int64_t kstep = (int64_t)KT * dp.krow;
    const int64_t vhead = (int64_t)D * p.ldvt * 2;     // bytes between two heads' V^T rows

    // ---- the first ticket. The NEXT item is always drawn late — about eight tiles before the stream needs its first K tile — not a
    //      whole item ahead: a ticket held early is an item no idle CU can take (in the first build the tail of the 5B shape, 30 half
    //      pieces per XCD reserved an item early by CUs that still had a whole block to finish, cost 6 %) ----
    if (tid == 0) __atomic_store_n(&mail[0], draw_ticket(p, cnt, xcd, nwg), __ATOMIC_RELAXED);
    __syncthreads();
    Item cur = decode_item(p, __builtin_amdgcn_readfirstlane(__atomic_load_n(&mail[0], __ATOMIC_RELAXED)), nt);
    Item nxt = decode_item(p, -1, nt);
    __syncthreads();
    if (cur.nsp == 0) return;
    bool have_nxt = false;          // nxt is decoded
    bool drew = false;              // thread 0 has put a fresh ticket into the mailbox; the next tile's barrier publishes it

    Blk A, B;
    u32x4 ring[RD];
    const char* kg = nullptr;       // source of the next K tile to fetch (4 tiles ahead of the tile being computed)
    const char* vg = nullptr;       // ... of the next V^T tile (3 tiles ahead)
    int kleft = 0, vleft = 0;       // tiles of the stream's current item still to fetch
    int g = 0;                      // tiles computed since the last cold start: tile g lives in slot g & 3
    int t = 0;                      // key tile (of cur) the next tile step computes
    bool cold = true;
    bool first = false;             // the next tile step is the first behind an item boundary of the stream (KIND 3)
    const unsigned junk_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int*)junk + wave * 512;

    for (;;) {
        const char* const kb_cur = reinterpret_cast<const char*>(p.K + cur.h * D);
        const char* const vb_cur = reinterpret_cast<const char*>(p.Vt) + (int64_t)cur.h * vhead;
        const int q0 = p.q_lo + cur.qb * QB7 + wave * 64;
        if (cold) {
            // ---- cold start (the workgroup's first item; the item after a robust rerun): attn_fwd7's prologue on slots 0.. ----
            cold = false;
            dp.kbase = kb_cur;
            dp.vbase = vb_cur;
            load_q<OA, QA, true, true>(p, A, q0 + ql, cur.h, hi);
            load_q<OB, QB, true, true>(p, B, q0 + 32 + ql, cur.h, hi);
            __builtin_amdgcn_sched_barrier(0);
            const int n = cur.t1 - cur.t0;             // >= 5 (launcher)
            // K(t0) | K(t0+1) V(t0) | K(t0+2) V(t0+1) | K(t0+3) V(t0+2) -> slots 0, 1, 2, 3 / 0, 1, 2
            dma7_k(dp, p, cur.t0, false, cx.lds0, wave);
            dma7_k(dp, p, cur.t0 + 1, false, cx.lds0 + SLOT, wave);
            dma7_v(dp, p, cur.t0, false, cx.lds0 + VB, wave);
            dma7_k(dp, p, cur.t0 + 2, false, cx.lds0 + 2 * SLOT, wave);
            dma7_v(dp, p, cur.t0 + 1, false, cx.lds0 + VB + SLOT, wave);
            dma7_k(dp, p, cur.t0 + 3, false, cx.lds0 + 3 * SLOT, wave);
            dma7_v(dp, p, cur.t0 + 2, false, cx.lds0 + VB + 2 * SLOT, wave);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");      // K(t0) has landed
            __builtin_amdgcn_s_barrier();
            NOP_PAD();
            fill_kcache(cx, 0);
            phase<true, OA, QA, OB, true, false, false, false, false, false, false>(cx, A, B, ring, 0, 0, 0, dp, nullptr, nullptr);
            phase<true, OB, QB, OA, true, false, false, false, true, false, false>(cx, B, A, ring, 0, 0, 0, dp, nullptr, nullptr);
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");      // K(t0+1) has landed
            __builtin_amdgcn_s_barrier();
            fill_kcache(cx, SLOT);                                   // the steady tile finds K(t+1) in the cache
            g = 0;
            t = cur.t0;
            kg = kb_cur + (int64_t)(cur.t0 + 4) * kstep;
            vg = vb_cur + (int64_t)(cur.t0 + 3) * (KT * 2);
            kleft = n - 4;
            vleft = n - 3;
            // Everything hipcc itself has in flight ends HERE: the spill reloads on the edges that lead to a cold start (the first item, the
            // robust rerun). hipcc cannot count the LDS-DMA pieces, so a load of its own that is still pending when a tile first touches
            // the register costs an s_waitcnt vmcnt(0) INSIDE the tile (inside the steady loop, in the first build: one drain of the LDS-DMA
            // queue per trip). The builtin, unlike an asm statement, clears hipcc's scoreboard.
            __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0)
            first = false;
        }
        kg = uniform_ptr(kg);
        vg = uniform_ptr(vg);
        kleft = __builtin_amdgcn_readfirstlane(kleft);
        vleft = __builtin_amdgcn_readfirstlane(vleft);
        g = __builtin_amdgcn_readfirstlane(g);
        t = __builtin_amdgcn_readfirstlane(t);
        int rem = cur.t1 - t;
        bool touched = false;

        // the next item's ticket: drawn by thread 0 when at most 8 tiles of this item are left, read by everybody one tile (one barrier) later
        auto next_ticket = [&]() {
            if (have_nxt) return;
            if (drew) {
                nxt = decode_item(p, __builtin_amdgcn_readfirstlane(__atomic_load_n(&mail[0], __ATOMIC_RELAXED)), nt);
                have_nxt = true;
                drew = false;
            } else if (rem <= 8) {
                if (tid == 0) __atomic_store_n(&mail[0], draw_ticket(p, cnt, xcd, nwg), __ATOMIC_RELAXED);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                drew = true;
            }
        };
        // (items are at least 5 tiles long, so the ticket is known before the stream wraps; should it ever not be — draw and publish now)
        auto need_nxt = [&]() {
            if (have_nxt) return;
            if (!drew && tid == 0) __atomic_store_n(&mail[0], draw_ticket(p, cnt, xcd, nwg), __ATOMIC_RELAXED);
            __syncthreads();
            nxt = decode_item(p, __builtin_amdgcn_readfirstlane(__atomic_load_n(&mail[0], __ATOMIC_RELAXED)), nt);
            __syncthreads();
            have_nxt = true;
            drew = false;
        };
        // the stream leaves an item behind its last tile: on to the next item's first tile, or — nothing left — to a harmless re-fetch
        // of this item's first tile (the statements stay unconditional; nobody reads what they bring)
        auto wraps = [&]() {
            if (kleft == 0) {
                need_nxt();
                if (nxt.nsp) {
                    kg = reinterpret_cast<const char*>(p.K + nxt.h * D) + (int64_t)nxt.t0 * kstep;
                    kleft = nxt.t1 - nxt.t0;
                } else {
                    kg = kb_cur + (int64_t)cur.t0 * kstep;
                    kleft = 1 << 28;
                }
            }
            if (vleft == 0) {
                need_nxt();
                if (nxt.nsp) {
                    vg = reinterpret_cast<const char*>(p.Vt) + (int64_t)nxt.h * vhead + (int64_t)nxt.t0 * (KT * 2);
                    vleft = nxt.t1 - nxt.t0;
                } else {
                    vg = vb_cur + (int64_t)cur.t0 * (KT * 2);
                    vleft = 1 << 28;
                }
            }
        };
        // Q' touch: the next item's query rows are pulled towards the L2 a few tiles before bubble 1 loads them into the AGPRs — one dword
        // of each 128-byte half row per lane, by LDS-DMA into a junk area (no destination register that the late data could clobber). Two
        // more pieces in the queue: the next counted wait is that much stricter, nothing else.
        auto touch = [&]() {
            touched = true;
            const unsigned l = fresh_lane();       // (not the kernel's long-lived lane values: those sit in scratch by now, and their reload would wait)
            const int qn = p.q_lo + nxt.qb * QB7 + wave * 64 + (int)(l & 31);
            const char* qbase = reinterpret_cast<const char*>(p.Q + nxt.h * D);
            const int qa = qn < p.Lq ? qn : p.Lq - 1, qb2 = qn + 32 < p.Lq ? qn + 32 : p.Lq - 1;
            glds4(qbase, (unsigned)qa * (unsigned)(p.ldq * 2) + (l >> 5) * 128u, junk_lds);
            glds4(qbase, (unsigned)qb2 * (unsigned)(p.ldq * 2) + (l >> 5) * 128u, junk_lds + 256);
        };
        auto stepped = [&]() {
            ++g;
            ++t;
            --rem;
            --kleft;
            --vleft;
        };

        next_ticket();
        wraps();
        if (g > 0) TRACE_STAMP(2);
        // ---- all tiles of the item but its last two: steady code ----
        if (first) {                                     // (n >= 5: rem > 2 here)
            first = false;
            tile8_any<3>(g, cx, dp, kg, vg, kstep, A, B, ring, 0);
            stepped();
            next_ticket();
            wraps();
        }
        while (rem > 2) {
            if (rem <= 5 && !touched && have_nxt && nxt.nsp) touch();
            if ((g & 3) == 1 && rem >= 10 && kleft >= 4 && vleft >= 4) {
                // (as attn_fwd7: claims dead score registers so that a spill reload parked in them is waited for HERE, not inside the loop.
                // Only s[0]: element [1][15] of the scores is still read by the softmax drain in the first gap of the next tile.)
                asm volatile("" : "=v"(A.s[0]));
#pragma unroll 1
                do {
                    steady7<1, true>(cx, dp, kg, vg, kstep, A, B, ring);
                    steady7<2, true>(cx, dp, kg, vg, kstep, A, B, ring);
                    steady7<3, true>(cx, dp, kg, vg, kstep, A, B, ring);
                    steady7<0, true>(cx, dp, kg, vg, kstep, A, B, ring);
                    g += 4;
                    t += 4;
                    rem -= 4;
                    kleft -= 4;
                    vleft -= 4;
                } while (rem >= 10 && kleft >= 4 && vleft >= 4);      // (the trips end where the next ticket is due: rem <= 9)
            } else {
                tile8_any<0>(g, cx, dp, kg, vg, kstep, A, B, ring, 0);
                stepped();
            }
            next_ticket();
            wraps();
        }
        need_nxt();
        if (!touched && nxt.nsp) touch();
        const int jl = (cur.t1 - 1) * KT;                // first key of the item's last tile
        // ---- the tile before the last ----
        TRACE_STAMP(3);      // (experiment builds, trace.hpp; the stamps of a workgroup's LAST item boundary survive: tools/trace8.py)
        tile8_any<1>(g, cx, dp, kg, vg, kstep, A, B, ring, jl);
        TRACE_STAMP(4);
        stepped();
        wraps();
        if (nxt.nsp) {
            // ---- bubble 1: Q' into the AGPRs Q has just left (its last use was S(last) in the tile above) ----
            const unsigned l = fresh_lane();
            const int qn = p.q_lo + nxt.qb * QB7 + wave * 64 + (int)(l & 31);
            const char* qbase = reinterpret_cast<const char*>(p.Q + nxt.h * D);
            const int qa = qn < p.Lq ? qn : p.Lq - 1, qb2 = qn + 32 < p.Lq ? qn + 32 : p.Lq - 1;
            load_q_agpr<QA>(qbase, (unsigned)qa * (unsigned)(p.ldq * 2) + (l >> 5) * 16u);
            load_q_agpr<QB>(qbase, (unsigned)qb2 * (unsigned)(p.ldq * 2) + (l >> 5) * 16u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TRACE_STAMP(5);
            // ---- the last tile, its score half already the next item's ----
            tile8_any<2>(g, cx, dp, kg, vg, kstep, A, B, ring, jl);
            TRACE_STAMP(6);
            ++g;
            --kleft;
            --vleft;
        } else {
            // ---- the last tile of the workgroup's last item ----
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int vb = (g & 3) * SLOT;
            const Ctx cm = masked_ctx(cx, jl - cx.Lk);
            phase<true, OA, QA, OB, false, true, true, true, true, true, false>(cm, A, B, ring, vb, 0, 0, dp, nullptr, nullptr);
            phase<true, OB, QB, OA, false, true, true, true, false, true, false>(cm, B, A, ring, vb, 0, 0, dp, nullptr, nullptr);
        }
        NOP_PAD();                                       // pending MFMA results -> accvgpr reads

        // ---- bubble 2: the item's range vote, its O^T out, the accumulators back to zero ----
        const bool ok = block_in_range<OA>(A) & block_in_range<OB>(B);
        const int wave_ok = __all(ok) ? 1 : 0;
        if (lane == 0) votes[wave] = wave_ok;
        __syncthreads();
        const int all_ok = votes[0] & votes[1] & votes[2] & votes[3];
        __syncthreads();
        if (nxt.nsp) TRACE_STAMP(1);
        if (__builtin_expect(!__builtin_amdgcn_readfirstlane(all_ok), 0)) {
            // out of the base-free body's range: the whole item again, cold, on attn_fwd7's rescaling pieces (any pointers, any Lk); the
            // stream's prefetched tiles are lost, the next item starts cold
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            dp.kbase = kb_cur;
            dp.vbase = vb_cur;
            run_keys<false, true>(p, cx, dp, A, B, q0, ql, cur.h, hi, cur.t0, cur.t1, nt, tid);
            cold = true;

        }
        // hipcc's own loads end HERE, in front of the O^T stores: the spill reloads of what it parked across the boundary tile, the rerun's.
        // Behind the stores nothing of hipcc's may be pending when the next tile starts — it would wait for it with a count that also
        // covers the stores (the builtin, unlike an asm statement, clears hipcc's scoreboard; what it waits for besides is the boundary
        // tile's 8 pieces, 3 us old).
        __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
        if (cur.nsp > 1) {
            const int row0 = p.q_lo + p.tail_qb * QB7;
            const int64_t rows = p.Lq - row0;
            store_partial8<OA>(p, A, q0 + ql, cur.h, hi, cur.sp, rows, row0);
            store_partial8<OB>(p, B, q0 + 32 + ql, cur.h, hi, cur.sp, rows, row0);
        } else {
            store_block8<OA>(p, A, q0 + ql, cur.h, hi);
            store_block8<OB>(p, B, q0 + 32 + ql, cur.h, hi);
        }
        if (nxt.nsp == 0) break;
        cur = nxt;
        have_nxt = false;
        drew = false;
        t = cur.t0;                                      // stream mode: S(t0) and half of its softmax exist already; the next tile step is "tile t0"
        if (!cold) {
            for_regs<OA, 64>([&](auto r) { agpr_set<decltype(r)::value>(0u); });
            for_regs<OB, 64>([&](auto r) { agpr_set<decltype(r)::value>(0u); });
            A.z.l_run = 0.f;
            B.z.l_run = 0.f;
            NOP_PAD();                                   // accvgpr writes -> MFMA C operands
            first = true;
        }
    }
}

}  // namespace

void yume_attn8_launch(const AttnArgs& a, int* counters, int nwg, hipStream_t st) {
    hipLaunchKernelGGL(attn_fwd_kernel_v8, dim3((unsigned)nwg), dim3(256), 0, st, a, counters, nwg);
}
