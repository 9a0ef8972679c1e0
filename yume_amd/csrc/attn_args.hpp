// attn_args.hpp — launch arguments shared by the attention kernels (attn_fwd.hip, attn_fwd7.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

struct AttnArgs {
    const unsigned short* Q; int64_t ldq;
    const unsigned short* K; int64_t ldk;
    const unsigned short* Vt; int64_t ldvt;
    unsigned short* O; int64_t ldo;
    int Lq, Lk, H;
    float scale_log2;  // softmax scale * log2(e); exactly 1 when q_prescaled
    int q_prescaled;   // Q already carries softmax scale * log2(e) (YUME_ATTN_Q_PRESCALED): the scores are the exponents
    int accumulate;
    int nqb;           // query blocks per head (of the rows [q_lo, Lq) this launch covers)
    int q_lo;          // first query row of this launch
    // key-range split (gridDim.y = splits > 1, v2 kernel only): split s walks key tiles [nt*s/splits, nt*(s+1)/splits) and writes
    // its UNNORMALISED O (fp32) + running max + row sum here; attn_combine_kernel merges the splits
    float* part_o;     // [splits, Lq - q_lo, H*128]
    float* part_ml;    // [splits, Lq - q_lo, H, 2]
    // v7 kernel only: query blocks >= tail_qb are cut into `splits` key ranges inside the same launch (their pieces are dispatched
    // after the whole blocks and fill the partial last round of workgroups); partials as above with rows = Lq - (q_lo + 256*tail_qb)
    int tail_qb, splits;
};

// attn_fwd7.hip: 4-wave / 64-queries-per-wave kernel (one wave per SIMD, 512 registers); grid = ceil(H/8) * nqb * 8 blocks
void yume_attn7_launch(const AttnArgs& a, hipStream_t st);
// attn_fwd8.hip: the same kernel as `nwg` persistent workgroups over one continuous K / V^T stream; items (the query blocks of a.nqb /
// a.tail_qb / a.splits, as for attn_fwd7) are drawn by ticket from `counters`, one 64-byte set of the caller's counter workspace
// (counters.hpp). Requires a.q_prescaled, K readable and V^T finite up to a whole number of 64-key tiles, every item >= 5 key tiles.
void yume_attn8_launch(const AttnArgs& a, int* counters, int nwg, hipStream_t st);
