// gemm_core.hpp — the bf16 MFMA GEMM pipeline shared by the dense GEMM (gemm_bf16.hip) and the implicit-GEMM
// causal convolutions of the VAE (conv3d.hip):   acc[m,n] = sum_k A[m,k] * W[n,k].
//
// Two kernels share the operand path (both operands K-contiguous in 16-byte chunks, HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 16 B/lane); the LDS image is lane-linear, the bank-conflict swizzle is applied on the SOURCE
// address (16-byte chunk c of row r is fetched from logical chunk c ^ (r & 7)) and undone on the ds_read_b128):
//   gemm128_kernel  tile 128 x 128 x 64, 256 threads = 4 waves (2 x 2), each wave 64 x 64 = 4 x 4 v_mfma_f32_16x16x32_bf16
//                   accumulators, double buffered, one barrier per K tile, two workgroups per CU. Small / ragged problems,
//                   batched launches (gridDim.y), and the epilogues that need an LDS restage.
//   gemm256_kernel  tile 256 x 256 x 64, 512 threads = 8 waves (2 x 4), each wave 64 x 128 as [mh][nh] quarter tiles of 4 x 2
//                   accumulators; half-tiles are re-staged phase by phase (4 phases per K tile), the two waves sharing a SIMD
//                   run one barrier apart (MODE 2) so one issues MFMAs while the other issues its reads. One workgroup per CU;
//                   chosen automatically once the problem fills the chip (launch_auto). The product is computed transposed
//                   (SWAP) for row-major outputs so a lane holds 4 consecutive columns and stores 8/16-byte vectors straight
//                   from the accumulators; whole tiles take epilogue_rows_full (all loads of 8 accumulator groups in flight
//                   before the first use), edge tiles the guarded per-group form.
// The A operand is produced by an `ALoad` policy: a row-major matrix (PlainA) or a gather from a channels-last [T,H,W,C]
// activation with zero padding / causal frame cache / folded 2x upsample (conv3d.hip).
// Fused epilogues: bias / GELU (tanh, erf) / GEGLU / gate*y + fp32 residual / + bf16 addend / transposed (K-major V^T) /
// frame-interleaved stores.
// Workgroup order: bijective XCD remap (block b runs on XCD b % 8) + grouped traversal (group_m M-tiles) so the blocks
// resident on one XCD share A row-panels and W column-panels in that XCD's L2.
#pragma once
#include "common.hpp"
#include "trace.hpp"
#include <stdlib.h>
#include <type_traits>

namespace gemm_core {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHR = 256;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;       // double buffered = 64 KiB

// internal epilogue ids beyond the public YUME_EPI_* (conv3d.hip)
constexpr int EPI_BF16_ADD = 16;      // out bf16 = acc + bias + add[m, n]   (add bf16, ld = ldadd)
constexpr int EPI_BF16_TSPLIT = 17;
// public YUME_EPI_BF16_GEGLU = 6: W rows interleaved (gate_j, fc1_j): out bf16 [M, N/2], out[m, j] = acc[2j+1] * gelu_tanh(acc[2j])   // out bf16 row m=(t,hw), col n=(j,c) -> out[((2t+j)*HW + hw), c]   (time_conv interleave)

struct Epilogue {
    const float* bias;
    void* out; int64_t ldo;
    const float* gate; int64_t gate_stride; const int32_t* row_idx;
    unsigned short* outT; int64_t ldt; int n_split;
    const unsigned short* add; int64_t ldadd;
    int hw;   // TSPLIT: positions per frame
};

struct Problem {
    const unsigned short* W; int64_t ldw;
    int M, N, K;
    int tiles_m, tiles_n;
    int group_m;   // 256^2 kernel: M-tiles per traversal group (L2 reuse knob)
    int64_t bsW, bsO;   // batched launch (gridDim.y > 1, 128^2 kernel): W stride in elements, out stride in BYTES per batch index
    int epi_direct;     // A/B switch (env YUME_GEMM_EPI_DIRECT=1): 256^2 kernel stores bf16 tiles straight from the accumulators
    // split-K tail of gemm_w4_kernel (gemm_w4.hpp, r6): the first sk_dp tiles of the order are whole-tile workgroups; each of the last sk_tiles
    // tiles is cut along K into sk_s slices — slice 0 = K tiles [0, sk_lh) (its workgroup finishes the tile), slices 1 .. sk_s - 1 of sk_lt K
    // tiles each (the last takes the rest) — run by sk_wgs more workgroups, the tail slices sk_q per workgroup. sk_ws = caller-owned scratch
    // (one partial tile and one flag per tail slice). sk_wgs = 0: off.
    int sk_dp, sk_tiles, sk_wgs, sk_s, sk_q, sk_lh, sk_lt;
    char* sk_ws;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

// ---- A loaders -------------------------------------------------------------------------------------------
// contract: init(m0, tid) once; then for kt = 0,1,2,... in order: src(rr, kt) for rr = 0..3 returns the global
// address of the 16-byte chunk that belongs at LDS position (row = rr*32 + tid/8, chunk = tid%8), i.e. logical
// k-chunk (tid%8) ^ (row & 7) of K tile kt; advance() after each tile.
struct PlainA {
    const unsigned short* A; int64_t lda; int M;
    int64_t bsA;   // batched launch: A stride in elements per batch index
    const unsigned short* rowp[4];
    __device__ __forceinline__ void batch_offset(int b) { A += (int64_t)b * bsA; }
    // kshift selects the LDS swizzle key of a row: (row >> kshift) & 7 (0 for the 16x16 fragments, 1 for 32x32 ones)
    __device__ __forceinline__ void init(int m0, int tid, int rpr = 32, int kshift = 0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = rr * rpr + (tid >> 3);
            int gr = m0 + r;
            gr = gr < M ? gr : M - 1;
            rowp[rr] = A + (int64_t)gr * lda + (((tid & 7) ^ ((r >> kshift) & 7)) << 3);
        }
    }
    __device__ __forceinline__ const unsigned short* src(int rr, int kt) const { return rowp[rr] + kt * BK; }
    __device__ __forceinline__ void advance() {}
    // element offset inside a W row of K tile kt (a loader may walk K in its own order: conv3d.hip)
    __device__ __forceinline__ int wk(int kt, bool /*behind*/) const { return kt * BK; }
};

__device__ __forceinline__ void stage_b(const unsigned short* __restrict__ W, int64_t ldw, int n0, int N, int k0,
                                        char* lds_tile, int tid, int wave) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = rr * 32 + (tid >> 3);
        const int gc = (tid & 7) ^ (r & 7);
        int gr = n0 + r;
        gr = gr < N ? gr : N - 1;
        const unsigned short* g = W + (int64_t)gr * ldw + k0 + gc * 8;
        char* l = lds_tile + rr * 4096 + wave * 1024;  // wave-uniform base; hardware adds lane*16
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)g, (lds_void*)l, 16, 0, 0);
    }
}

template <class ALoad>
__device__ __forceinline__ void stage_a(ALoad& al, int kt, char* lds_tile, int wave) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const unsigned short* g = al.src(rr, kt);
        char* l = lds_tile + rr * 4096 + wave * 1024;
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)g, (lds_void*)l, 16, 0, 0);
    }
    al.advance();
}

__device__ __forceinline__ bf16x8_t lds_frag(const char* tile, int row, int chunk) {
    const char* p = tile + row * 128 + ((chunk ^ (row & 7)) << 4);
    return *reinterpret_cast<const bf16x8_t*>(p);
}

// ---- epilogue helpers (shared by the 128^2 and 256^2 kernels) ------------------------------------------------
// A wave restages a 64x64 fp32 sub-tile through its private 16 KiB of LDS. Staging row r <-> global row m_base + r;
// staging columns [0,32) <-> global n_base0 + c, [32,64) <-> n_base1 + (c - 32).
__device__ __forceinline__ void stage_acc(float* ep, const f32x4& a, int i /*16-row block*/, int j /*16-col block*/,
                                          int lane, bool transposed) {
    if (!transposed) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ep[(i * 16 + 4 * (lane >> 4) + r) * 64 + j * 16 + (lane & 15)] = a[r];
    } else {
        // [n][m] image, 4-float granule g of row n stored at granule g ^ (n & 15)
        const int n = j * 16 + (lane & 15);
        const int g = i * 4 + (lane >> 4);
        *reinterpret_cast<f32x4*>(ep + n * 64 + ((g ^ (n & 15)) << 2)) = a;
    }
}

template <int EPI>
__device__ __forceinline__ void wave_epilogue(const float* ep, bool transposed, const Problem& p, const Epilogue& e,
                                              int lane, int m_base, int n_base0, int n_base1) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own LDS writes done (region is private to the wave)
    __builtin_amdgcn_wave_barrier();
    const int sub = lane >> 4;         // row within a pass of 4
    const int c4 = (lane & 15) << 2;   // first of 4 contiguous staging columns
    if (!transposed) {
        const int n = (c4 < 32 ? n_base0 : n_base1 - 32) + c4;
        f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (e.bias && n < p.N) bias4 = *reinterpret_cast<const f32x4*>(e.bias + n);
#pragma unroll 4
        for (int ps = 0; ps < 16; ++ps) {
            const int rl = ps * 4 + sub;
            const int m = m_base + rl;
            if (m >= p.M || n >= p.N) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(ep + rl * 64 + c4);
            v += bias4;
            if (EPI == YUME_EPI_BF16 || EPI == YUME_EPI_BF16_GELU || EPI == YUME_EPI_BF16_SPLITT ||
                EPI == YUME_EPI_BF16_GELU_ERF || EPI == EPI_BF16_ADD || EPI == EPI_BF16_TSPLIT) {
                if (EPI == YUME_EPI_BF16_GELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = gelu_tanh(v[q]);
                }
                if (EPI == YUME_EPI_BF16_GELU_ERF) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = 0.5f * v[q] * (1.0f + erff(v[q] * 0.7071067811865476f));
                }
                if (EPI == EPI_BF16_ADD) {
                    const u32x2 a2 = *reinterpret_cast<const u32x2*>(e.add + (int64_t)m * e.ldadd + n);
                    v[0] += bf16_to_f32((unsigned short)(a2[0] & 0xffffu));
                    v[1] += bf16_to_f32((unsigned short)(a2[0] >> 16));
                    v[2] += bf16_to_f32((unsigned short)(a2[1] & 0xffffu));
                    v[3] += bf16_to_f32((unsigned short)(a2[1] >> 16));
                }
                u32x2 o;
                o[0] = pack_bf16x2(v[0], v[1]);
                o[1] = pack_bf16x2(v[2], v[3]);
                int64_t orow = m;
                int ocol = n;
                if (EPI == EPI_BF16_TSPLIT) {
                    const int ch = p.N >> 1;
                    const int j = n >= ch ? 1 : 0;
                    const int t = m / e.hw;
                    orow = (int64_t)m + (int64_t)(t + j) * e.hw;   // ((2t + j) * hw + (m - t*hw))
                    ocol = n - j * ch;
                }
                *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(e.out) + orow * e.ldo + ocol) = o;
            } else if (EPI == YUME_EPI_F32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(e.out) + (int64_t)m * e.ldo + n) = v;
            } else {  // YUME_EPI_RESID
                float* xo = reinterpret_cast<float*>(e.out) + (int64_t)m * e.ldo + n;
                f32x4 x = *reinterpret_cast<const f32x4*>(xo);
                if (e.gate) {
                    const int64_t row = e.row_idx ? (int64_t)e.row_idx[m] : 0;
                    const f32x4 g = *reinterpret_cast<const f32x4*>(e.gate + row * e.gate_stride + n);
                    x += v * g;
                } else {
                    x += v;
                }
                *reinterpret_cast<f32x4*>(xo) = x;
            }
        }
    } else {
        // rows of the image are output features (staging n), columns are tokens m
        const int m = m_base + c4;
#pragma unroll 4
        for (int ps = 0; ps < 16; ++ps) {
            const int nl = ps * 4 + sub;
            const int n = (nl < 32 ? n_base0 : n_base1 - 32) + nl;
            if (n >= p.N || m >= p.M) continue;
            const int g = lane & 15;
            f32x4 v = *reinterpret_cast<const f32x4*>(ep + nl * 64 + ((g ^ (nl & 15)) << 2));
            const float bn = e.bias ? e.bias[n] : 0.f;
            unsigned short* dst = e.outT + (int64_t)(n - e.n_split) * e.ldt + m;
            if (m + 3 < p.M) {
                u32x2 o;
                o[0] = pack_bf16x2(v[0] + bn, v[1] + bn);
                o[1] = pack_bf16x2(v[2] + bn, v[3] + bn);
                *reinterpret_cast<u32x2*>(dst) = o;
            } else {
                for (int q = 0; q < 4 && m + q < p.M; ++q) dst[q] = f32_to_bf16(v[q] + bn);
            }
        }
    }
}

template <int EPI, class ALoad>
__global__ __launch_bounds__(NTHR, 2) void gemm128_kernel(Problem p, ALoad al, Epilogue e) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    if (gridDim.y > 1) {          // batched launch: independent problems of one shape, operands a fixed stride apart
        al.batch_offset(blockIdx.y);
        p.W += (int64_t)blockIdx.y * p.bsW;
        e.out = reinterpret_cast<char*>(e.out) + (int64_t)blockIdx.y * p.bsO;
    }

    // ---- workgroup -> tile (XCD-aware, grouped) ----
    const int nwg = p.tiles_m * p.tiles_n;
    int wg;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    constexpr int GROUP_M = 8;
    const int width = GROUP_M * p.tiles_n;
    const int group = wg / width;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (wg % width) % gsz;
    const int tn = (wg % width) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    TRACE_STAMP(0);
    const int nk = p.K / BK;
    al.init(m0, tid);
    stage_a(al, 0, smem, wave);
    stage_b(p.W, p.ldw, n0, p.N, al.wk(0, true), smem + TILE_BYTES, tid, wave);       // (the loader has moved on to tile 1)

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed (own loads) ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... for every wave, and every wave is done reading the other buffer
        __syncthreads();
        if (kt + 1 < nk) {
            char* nb = smem + (cur ^ 1) * STAGE_BYTES;
            stage_a(al, kt + 1, nb, wave);
            stage_b(p.W, p.ldw, n0, p.N, al.wk(kt + 1, true), nb + TILE_BYTES, tid, wave);
        }
        const char* At = smem + cur * STAGE_BYTES;
        const char* Bt = At + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t a[4], b[4];
            const int ch = ks * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = lds_frag(At, wm * 64 + i * 16 + (lane & 15), ch);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = lds_frag(Bt, wn * 64 + j * 16 + (lane & 15), ch);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        cur ^= 1;
    }
    __syncthreads();  // all waves finished with the operand tiles; LDS is reused for the epilogue
    TRACE_STAMP(1);

    // ---- epilogue: per-wave 64x64 fp32 restage (16 KiB per wave) ----
    float* ep = reinterpret_cast<float*>(smem) + wave * (64 * 64);
    const int wm0 = m0 + wm * 64, wn0 = n0 + wn * 64;
    const bool transposed = (EPI == YUME_EPI_BF16_SPLITT) && (n0 >= e.n_split);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) stage_acc(ep, acc[i][j], i, j, lane, transposed);
    wave_epilogue<EPI>(ep, transposed, p, e, lane, wm0, wn0, wn0 + 32);
    TRACE_STAMP(2);
}

// =====================================================================================================================
// 256(M) x 256(N) x 64(K) tile, 512 threads = 8 waves (2 along M x 4 along N), one workgroup per CU, 128 KiB of LDS:
// two K-tile slots, each made of four 16 KiB HALF-TILES  A0 | A1 | B0 | B1  (128 rows x 64 k).
// A wave (wr, wc) owns rows {mh*128 + wr*64 + [0,64)} and columns {nh*128 + wc*32 + [0,32)}, mh, nh in {0,1}, so that its
// (mh, nh) output quadrant reads exactly half-tiles A_mh and B_nh. A K tile is computed in four phases
//      P1 (0,0): read A0, B0      P2 (0,1): read B1      P3 (1,1): read A1      P4 (1,0): read B0
// (16 MFMAs each), hence A0 is dead after P1, B1 after P2, A1 after P3, B0 after P4, and each phase re-stages the half-tile
// that died one phase earlier with the data of two K tiles ahead:
//      (t,P1): B0(t+1)     (t,P2): A0(t+2)     (t,P3): B1(t+2)     (t,P4): A1(t+2)
// One s_barrier per phase (after the phase's ds_reads have landed) is what makes the re-staging safe (WAR); the data
// hazard (RAW) is covered by ONE counted wait per K tile: at (t,P4) `s_waitcnt vmcnt(6)` leaves exactly the three newest
// half-tiles A0/B1/A1(t+2) in flight and retires everything K tile t+1 needs, and the P4 barrier publishes it to the
// other waves before anyone reads it in (t+1,P1). Loads therefore stay in flight across barriers for a whole K tile of
// MFMA work (64 MFMAs per wave) — the LDS-DMA equivalent of a 3-deep cp.async pipeline.
constexpr int NTHR256 = 512;
constexpr int HALF_BYTES = 128 * BK * 2;        // 16 KiB
constexpr int SLOT_BYTES = 4 * HALF_BYTES;      // 64 KiB
constexpr int LDS256_BYTES = 2 * SLOT_BYTES;    // 128 KiB

template <class ALoad>
__device__ __forceinline__ void stage_half_a(ALoad& al, int mh, int kt, char* half, int wave) {
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
        const unsigned short* g = al.src(mh * 2 + r2, kt);
        char* l = half + r2 * 8192 + wave * 1024;
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)g, (lds_void*)l, 16, 0, 0);
    }
}

__device__ __forceinline__ void stage_half_b(const unsigned short* const (&wrow)[4], int nh, int koff, char* half, int wave) {
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
        const unsigned short* g = wrow[nh * 2 + r2] + koff;
        char* l = half + r2 * 8192 + wave * 1024;
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)g, (lds_void*)l, 16, 0, 0);
    }
}

#define YUME_PHASE_SYNC()                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                           \
    __builtin_amdgcn_sched_barrier(0);                      \
    if (MODE == 1 || MODE == 2) __builtin_amdgcn_s_setprio(1)
#define YUME_PHASE_END()                                         \
    if (MODE == 1 || MODE == 2) __builtin_amdgcn_s_setprio(0);   \
    if (MODE >= 2) {                                             \
        asm volatile("" ::: "memory");                           \
        __builtin_amdgcn_s_barrier();                            \
        __builtin_amdgcn_sched_barrier(0);                       \
    }

// ---- vector epilogue stores straight from accumulators (256^2 kernels) ----
// v = C[m][n .. n+3] (row-major destinations)
template <int EPI>
__device__ __forceinline__ void store_row4(f32x4 v, int m, int n, const Problem& p, const Epilogue& e) {
    if (e.bias) v += *reinterpret_cast<const f32x4*>(e.bias + n);
    if (EPI == YUME_EPI_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(e.out) + (int64_t)m * e.ldo + n) = v;
    } else if (EPI == YUME_EPI_BF16_GEGLU) {
        const unsigned o = pack_bf16x2(v[1] * gelu_tanh(v[0]), v[3] * gelu_tanh(v[2]));
        *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(e.out) + (int64_t)m * e.ldo + (n >> 1)) = o;
    } else if (EPI == YUME_EPI_RESID) {
        float* xo = reinterpret_cast<float*>(e.out) + (int64_t)m * e.ldo + n;
        f32x4 x = *reinterpret_cast<const f32x4*>(xo);
        if (e.gate) {
            const int64_t row = e.row_idx ? (int64_t)e.row_idx[m] : 0;
            x += v * *reinterpret_cast<const f32x4*>(e.gate + row * e.gate_stride + n);
        } else {
            x += v;
        }
        *reinterpret_cast<f32x4*>(xo) = x;
    } else {
        if (EPI == YUME_EPI_BF16_GELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = gelu_tanh(v[q]);
        }
        if (EPI == YUME_EPI_BF16_GELU_ERF) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = 0.5f * v[q] * (1.0f + erff(v[q] * 0.7071067811865476f));
        }
        if (EPI == EPI_BF16_ADD) {
            const u32x2 a2 = *reinterpret_cast<const u32x2*>(e.add + (int64_t)m * e.ldadd + n);
            v[0] += bf16_to_f32((unsigned short)(a2[0] & 0xffffu));
            v[1] += bf16_to_f32((unsigned short)(a2[0] >> 16));
            v[2] += bf16_to_f32((unsigned short)(a2[1] & 0xffffu));
            v[3] += bf16_to_f32((unsigned short)(a2[1] >> 16));
        }
        int64_t orow = m;
        int ocol = n;
        if (EPI == EPI_BF16_TSPLIT) {
            const int ch = p.N >> 1;
            const int j = n >= ch ? 1 : 0;
            const int t = m / e.hw;
            orow = (int64_t)m + (int64_t)(t + j) * e.hw;
            ocol = n - j * ch;
        }
        u32x2 o;
        o[0] = pack_bf16x2(v[0], v[1]);
        o[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(e.out) + orow * e.ldo + ocol) = o;
    }
}
// v = C[m .. m+3][n] -> K-major outT[n - n_split][m .. m+3]
__device__ __forceinline__ void store_col4(f32x4 v, int m, int n, const Problem& p, const Epilogue& e) {
    const float bn = e.bias ? e.bias[n] : 0.f;
    unsigned short* dst = e.outT + (int64_t)(n - e.n_split) * e.ldt + m;
    if (m + 3 < p.M) {
        u32x2 o;
        o[0] = pack_bf16x2(v[0] + bn, v[1] + bn);
        o[1] = pack_bf16x2(v[2] + bn, v[3] + bn);
        *reinterpret_cast<u32x2*>(dst) = o;
    } else {
        for (int q = 0; q < 4 && m + q < p.M; ++q) dst[q] = f32_to_bf16(v[q] + bn);
    }
}

// Whole 256^2 tiles, row-major epilogues: a lane owns rows mlane + mh*128 + mi*16 and the 4-column groups nlane + nh*128 + ni*16.
// Every load the epilogue needs (bias, gate rows, the fp32 residual, the conv shortcut) is issued for a batch of 8 accumulator
// groups BEFORE the first dependent use, so a wave has 8-16 requests in flight instead of one (stores count in vmcnt on gfx9:
// the per-group form waited for the previous store's acknowledgement 32 times per tile).
template <int EPI>
__device__ __forceinline__ void epilogue_rows_full(const f32x4 (&acc)[2][2][4][2], int mlane, int nlane, const Problem& p, const Epilogue& e) {
    f32x4 b[2][2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            if (e.bias) b[nh][ni] = *reinterpret_cast<const f32x4*>(e.bias + nlane + nh * 128 + ni * 16);
            else b[nh][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    Epilogue e0 = e;
    e0.bias = nullptr;
    if (EPI == YUME_EPI_RESID) {
        int64_t grow[2][4];
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                grow[mh][mi] = (e.gate && e.row_idx) ? (int64_t)e.row_idx[mlane + mh * 128 + mi * 16] * e.gate_stride : 0;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
                f32x4 x[4][2], g[4][2];
                float* xo[4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    xo[mi] = reinterpret_cast<float*>(e.out) + (int64_t)(mlane + mh * 128 + mi * 16) * e.ldo + nlane + nh * 128;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) x[mi][ni] = *reinterpret_cast<const f32x4*>(xo[mi] + ni * 16);
                }
                if (e.gate) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            g[mi][ni] = *reinterpret_cast<const f32x4*>(e.gate + grow[mh][mi] + nlane + nh * 128 + ni * 16);
                }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const f32x4 v = acc[mh][nh][mi][ni] + b[nh][ni];
                        if (e.gate) x[mi][ni] += v * g[mi][ni];
                        else x[mi][ni] += v;
                        *reinterpret_cast<f32x4*>(xo[mi] + ni * 16) = x[mi][ni];
                    }
            }
    } else if (EPI == EPI_BF16_ADD) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
                u32x2 a2[4][2];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        a2[mi][ni] = *reinterpret_cast<const u32x2*>(e.add + (int64_t)(mlane + mh * 128 + mi * 16) * e.ldadd + nlane + nh * 128 + ni * 16);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        f32x4 v = acc[mh][nh][mi][ni] + b[nh][ni];
                        v[0] += bf16_to_f32((unsigned short)(a2[mi][ni][0] & 0xffffu));
                        v[1] += bf16_to_f32((unsigned short)(a2[mi][ni][0] >> 16));
                        v[2] += bf16_to_f32((unsigned short)(a2[mi][ni][1] & 0xffffu));
                        v[3] += bf16_to_f32((unsigned short)(a2[mi][ni][1] >> 16));
                        store_row4<YUME_EPI_BF16>(v, mlane + mh * 128 + mi * 16, nlane + nh * 128 + ni * 16, p, e0);
                    }
            }
    } else {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        store_row4<EPI>(acc[mh][nh][mi][ni] + b[nh][ni], mlane + mh * 128 + mi * 16, nlane + nh * 128 + ni * 16, p, e0);
    }
}

// MODE: 0 = one barrier per phase; 1 = + s_setprio(1) around the MFMA clusters; 2 = STAGGERED: a second barrier after every
// MFMA cluster and the wr = 1 half of the waves running one barrier behind the wr = 0 half, so that of the two waves
// sharing a SIMD (w and w+4) one is in its MFMA cluster while the other issues its ds_reads / LDS-DMA (+ setprio);
// 3 = staggered without setprio. WAR/RAW still hold: a half-tile is re-staged one full phase after its last reader's
// phase, and the counted vmcnt wait of every wave precedes the barrier that opens the first read of the next K tile.
// SWAP = true computes the transposed product (B fragment as the MFMA's first operand) so that a lane ends up with
// 4 CONSECUTIVE COLUMNS n of one row m: the row-major epilogues then store 8/16-byte vectors straight from the
// accumulators (no LDS restage). SWAP = false leaves 4 consecutive rows m per column n: the K-major V^T store.
typedef f32x4 Acc256[2][2][4][2];   // [mh][nh][mi][ni]: a wave's 64 x 128 part of the 256 x 256 tile

__device__ __forceinline__ void acc256_zero(Acc256& acc) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// K tiles [kb, kb + nk) of the tile at (m0, n0) accumulated ONTO acc. kb != 0 only with PlainA (stream-K segments).
template <class ALoad, int MODE, bool SWAP>
__device__ __forceinline__ void gemm256_mainloop(const Problem& p, ALoad& al, char* smem, int m0, int n0, int kb, int nk, Acc256& acc) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // W rows of this thread: half-tile nh, round r2 -> row nh*128 + r2*64 + tid/8 (clamped), source chunk swizzled
    const unsigned short* wrow[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = rr * 64 + (tid >> 3);
        int gr = n0 + r;
        gr = gr < p.N ? gr : p.N - 1;
        wrow[rr] = p.W + (int64_t)gr * p.ldw + (((tid & 7) ^ (r & 7)) << 3) + (int64_t)kb * BK;
    }
    al.init(m0, tid, 64);
    if constexpr (std::is_same<ALoad, PlainA>::value) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) al.rowp[rr] += (int64_t)kb * BK;
    }

    char* const s0 = smem;
    char* const s1 = smem + SLOT_BYTES;
    // half-tile offsets inside a slot
    constexpr int OA0 = 0, OA1 = HALF_BYTES, OB0 = 2 * HALF_BYTES, OB1 = 3 * HALF_BYTES;

    // ---- prologue: issue order = steady-state order (A0, B1, A1, B0 of tile 0; A0, B1, A1 of tile 1) ----
    // (W of K tile kt sits at element offset al.wk(kt, behind) of its row: `behind` = kt is the tile before the one the loader stands on)
    stage_half_a(al, 0, 0, s0 + OA0, wave);
    stage_half_b(wrow, 1, al.wk(0, false), s0 + OB1, wave);
    stage_half_a(al, 1, 0, s0 + OA1, wave);
    al.advance();
    stage_half_b(wrow, 0, al.wk(0, true), s0 + OB0, wave);
    if (nk > 1) {
        stage_half_a(al, 0, 1, s1 + OA0, wave);
        stage_half_b(wrow, 1, al.wk(1, false), s1 + OB1, wave);
        stage_half_a(al, 1, 1, s1 + OA1, wave);
        al.advance();
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (MODE >= 2 && wr == 1) {               // the wr = 1 waves run one barrier behind
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    const int arow = wr * 64 + (lane & 15);   // + mi*16 : row inside half-tile A_mh
    const int brow = wc * 32 + (lane & 15);   // + ni*16 : row inside half-tile B_nh
    const int lch = lane >> 4;                // + ks*4  : logical 16-byte chunk

    for (int t = 0; t < nk; ++t) {
        char* cs = (t & 1) ? s1 : s0;          // slot of K tile t (and t+2)
        char* ns = (t & 1) ? s0 : s1;          // slot of K tile t+1
        bf16x8_t a[4][2], b[2][2], b0[2][2];   // [mi][ks], [ni][ks]; b0 = the B0 fragments, kept from P1 for P4

        // ---------------- P1: quadrant (0,0) ----------------
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b0[ni][ks] = lds_frag(cs + OB0, brow + ni * 16, ks * 4 + lch);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a[mi][ks] = lds_frag(cs + OA0, arow + mi * 16, ks * 4 + lch);
        if (t + 1 < nk) stage_half_b(wrow, 0, al.wk(t + 1, true), ns + OB0, wave);
        YUME_PHASE_SYNC();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[0][0][mi][ni] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni][ks], a[mi][ks], acc[0][0][mi][ni], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi][ks], b0[ni][ks], acc[0][0][mi][ni], 0, 0, 0);
        YUME_PHASE_END();

        // ---------------- P2: quadrant (0,1) ----------------
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b[ni][ks] = lds_frag(cs + OB1, brow + ni * 16, ks * 4 + lch);
        if (t + 2 < nk) stage_half_a(al, 0, t + 2, cs + OA0, wave);
        YUME_PHASE_SYNC();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[0][1][mi][ni] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni][ks], a[mi][ks], acc[0][1][mi][ni], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi][ks], b[ni][ks], acc[0][1][mi][ni], 0, 0, 0);
        YUME_PHASE_END();

        // ---------------- P3: quadrant (1,1) ----------------
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a[mi][ks] = lds_frag(cs + OA1, arow + mi * 16, ks * 4 + lch);
        if (t + 2 < nk) stage_half_b(wrow, 1, al.wk(t + 2, false), cs + OB1, wave);
        YUME_PHASE_SYNC();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[1][1][mi][ni] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni][ks], a[mi][ks], acc[1][1][mi][ni], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi][ks], b[ni][ks], acc[1][1][mi][ni], 0, 0, 0);
        YUME_PHASE_END();

        // ---------------- P4: quadrant (1,0) ---------------- (B0 fragments still in registers from P1)
        if (t + 2 < nk) {
            stage_half_a(al, 1, t + 2, cs + OA1, wave);
            al.advance();
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // K tile t+1 complete; A0/B1/A1(t+2) stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        YUME_PHASE_SYNC();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[1][0][mi][ni] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni][ks], a[mi][ks], acc[1][0][mi][ni], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi][ks], b0[ni][ks], acc[1][0][mi][ni], 0, 0, 0);
        YUME_PHASE_END();
    }

    if (MODE >= 2 && wr == 0) __builtin_amdgcn_s_barrier();   // barrier counts of the two halves match again
}

// ---- epilogue: vector stores straight from the accumulators ----
// bf16 row-major outputs, tiles with all 256 columns inside N: the tile is restaged through the (now free) operand slots as a
// [256][256] bf16 image and leaves as whole 512-byte rows, 1 KiB per wave-instruction. Straight from the accumulators a lane owns
// 4 columns of one row, i.e. a store instruction writes sixteen 32-byte pieces of sixteen rows (measured: 5 % of a K = 3072 GEMM).
// 16-byte chunk c of row r sits at chunk c ^ (r & 31): the ds_write_b64 of 16 rows x one chunk column is 2-way conflicted (within the
// instruction's own transfer time), the ds_read_b128 of 32 chunks of one row is conflict-free.
template <int EPI>
__device__ __forceinline__ void epilogue_rows_lds(const Acc256& acc, char* smem, int m0, int n0, const Problem& p, const Epilogue& e) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, l4 = lane >> 4;
    f32x4 b[2][2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            if (e.bias) b[nh][ni] = *reinterpret_cast<const f32x4*>(e.bias + n0 + nh * 128 + wc * 32 + ni * 16 + 4 * l4);
            else b[nh][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    __syncthreads();                                  // every wave is done with the operand slots
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    f32x4 v = acc[mh][nh][mi][ni] + b[nh][ni];
                    if (EPI == YUME_EPI_BF16_GELU) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = gelu_tanh(v[q]);
                    }
                    if (EPI == YUME_EPI_BF16_GELU_ERF) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = 0.5f * v[q] * (1.0f + erff(v[q] * 0.7071067811865476f));
                    }
                    const int r = mh * 128 + wr * 64 + mi * 16 + l15;
                    const int c = nh * 128 + wc * 32 + ni * 16 + 4 * l4;          // bf16 column inside the tile
                    u32x2 o;
                    o[0] = pack_bf16x2(v[0], v[1]);
                    o[1] = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(smem + r * 512 + ((((c >> 3) ^ r) & 31) << 4) + ((c >> 2) & 1) * 8) = o;
                }
    __syncthreads();
    unsigned short* out = reinterpret_cast<unsigned short*>(e.out) + n0 + (lane & 31) * 8;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = wave * 32 + i * 2 + (lane >> 5);
        const u32x4 d = *reinterpret_cast<const u32x4*>(smem + r * 512 + ((((lane & 31) ^ r) & 31) << 4));
        if (m0 + r < p.M) *reinterpret_cast<u32x4*>(out + (int64_t)(m0 + r) * e.ldo) = d;
    }
}

template <int EPI, bool SWAP>
__device__ __forceinline__ void gemm256_epilogue(const Acc256& acc, const Problem& p, const Epilogue& e, int m0, int n0, char* smem) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, l4 = lane >> 4;
    if (SWAP && !p.epi_direct && n0 + 256 <= p.N && (e.ldo % 8) == 0 &&
        (EPI == YUME_EPI_BF16 || EPI == YUME_EPI_BF16_GELU || EPI == YUME_EPI_BF16_GELU_ERF || EPI == YUME_EPI_BF16_SPLITT)) {
        epilogue_rows_lds<EPI>(acc, smem, m0, n0, p, e);      // (workgroup-uniform condition)
        return;
    }
    if (SWAP && m0 + 256 <= p.M && n0 + 256 <= p.N) {         // whole tile (workgroup-uniform): batched loads, no guards
        epilogue_rows_full<EPI>(acc, m0 + wr * 64 + l15, n0 + wc * 32 + 4 * l4, p, e);
        return;
    }
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int mb = m0 + mh * 128 + wr * 64 + mi * 16;
                    const int nb = n0 + nh * 128 + wc * 32 + ni * 16;
                    f32x4 v = acc[mh][nh][mi][ni];
                    if (SWAP) {
                        const int m = mb + l15, n = nb + 4 * l4;        // v = C[m][n .. n+3]
                        if (m >= p.M || n >= p.N) continue;
                        store_row4<EPI>(v, m, n, p, e);
                    } else {
                        const int m = mb + 4 * l4, n = nb + l15;        // v = C[m .. m+3][n]  -> outT[n - n_split][m .. m+3]
                        if (m >= p.M || n >= p.N) continue;
                        store_col4(v, m, n, p, e);
                    }
                }
}

template <int EPI, class ALoad, int MODE, bool SWAP>
__device__ __forceinline__ void gemm256_body(const Problem& p, ALoad& al, const Epilogue& e, char* smem, int m0, int n0) {
    Acc256 acc;
    acc256_zero(acc);
    gemm256_mainloop<ALoad, MODE, SWAP>(p, al, smem, m0, n0, 0, p.K / BK, acc);
    gemm256_epilogue<EPI, SWAP>(acc, p, e, m0, n0, smem);
}

// (built, parity-green and measured slower, see profiles/r2_rejected_experiments.md: a v_mfma_f32_32x32x16_bf16 main loop (r1, -13 %); the
// fragment reads software-pipelined one phase ahead with ping-pong register sets (-1...-15 %); a one-wave-per-SIMD kernel with all
// 256 AGPRs as accumulators, K tiles of 32 and 0.5 ds_read per MFMA (-13...-20 %))

// position `wg` of the grouped tile order (group_m M-tiles x all N-tiles per group, M fastest) -> tile origin
__device__ __forceinline__ void tile_origin(const Problem& p, int wg, int& m0, int& n0) {
    const int width = p.group_m * p.tiles_n;
    const int group = wg / width;
    const int first_m = group * p.group_m;
    const int gsz = min(p.tiles_m - first_m, p.group_m);
    m0 = (first_m + (wg % width) % gsz) * 256;
    n0 = ((wg % width) / gsz) * 256;
}
// XCD x walks its own contiguous chunk [start, start + count) of that order (block b runs on XCD b % 8)
__host__ __device__ __forceinline__ void xcd_chunk(int nwg, int xcd, int& start, int& count) {
    const int q = nwg >> 3, r = nwg & 7;
    start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    count = q + (xcd < r ? 1 : 0);
}

template <int EPI, class ALoad, int MODE>
__global__ __launch_bounds__(NTHR256, 2) void gemm256_kernel(Problem p, ALoad al, Epilogue e) {
    __shared__ __attribute__((aligned(16))) char smem[LDS256_BYTES];
    // ---- workgroup -> tile (XCD-aware, grouped) ----
    int start, count, m0, n0;
    xcd_chunk(p.tiles_m * p.tiles_n, blockIdx.x & 7, start, count);
    tile_origin(p, start + (blockIdx.x >> 3), m0, n0);
    if (EPI == YUME_EPI_BF16_SPLITT && n0 >= e.n_split) gemm256_body<EPI, ALoad, MODE, false>(p, al, e, smem, m0, n0);
    else gemm256_body<EPI, ALoad, MODE, true>(p, al, e, smem, m0, n0);
}

// schedule variant of the 256^2 kernel (see MODE above); env YUME_GEMM_MODE overrides for A/B runs
inline int read_mode_env() {
    const char* v = getenv("YUME_GEMM_MODE");
    const int m = v ? atoi(v) : -1;
    return (m >= 0 && m <= 3) ? m : -1;
}
static const int g_mode256 = read_mode_env();   // -1: use the caller's default
inline int read_group_env() {
    const char* v = getenv("YUME_GEMM_GROUPM");
    const int g = v ? atoi(v) : 8;
    return g > 0 ? g : 8;
}
static const int g_group_m = read_group_env();

template <int EPI, class ALoad>
int launch256(const Problem& p128, const ALoad& al, const Epilogue& e, hipStream_t st, const char* what, int mode_default = 2) {
    Problem p = p128;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    p.group_m = g_group_m;
    { static const int d = [] { const char* v = getenv("YUME_GEMM_EPI_DIRECT"); return v ? atoi(v) : 0; }(); p.epi_direct = d; }
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(NTHR256);
    const int mode = g_mode256 >= 0 ? g_mode256 : mode_default;
    switch (mode) {
        case 0: hipLaunchKernelGGL((gemm256_kernel<EPI, ALoad, 0>), grid, block, 0, st, p, al, e); break;
        case 1: hipLaunchKernelGGL((gemm256_kernel<EPI, ALoad, 1>), grid, block, 0, st, p, al, e); break;
        case 3: hipLaunchKernelGGL((gemm256_kernel<EPI, ALoad, 3>), grid, block, 0, st, p, al, e); break;
        default: hipLaunchKernelGGL((gemm256_kernel<EPI, ALoad, 2>), grid, block, 0, st, p, al, e); break;
    }
    YUME_CHECK_LAUNCH(what);
    return YUME_OK;
}

// kernel selection: variant 1 = 128^2 tile, 2 = 256^2 tile, 0 = automatic (256^2 once it fills the chip)
// worth: what the 256^2 pipeline is worth per PADDED flop against the 128^2 kernel — 1.25 for the dense GEMM and the 8-wave kernel (measured
// 1.15-1.3 PF vs 0.85-1.0 PF); the convolution's one-wave-per-SIMD pipeline asks with 2.0 (r5: conv_w4 1.3 PF against 0.61 PF of the
// gathering 128^2 kernel on the same 384-channel launches, profiles/r5_vae21_decode_rocprofv3_kernel_stats.csv)
inline bool use_256(const Problem& p, int variant, bool split_ok, double worth = 1.25) {
    if (variant == 1 || !split_ok) return false;
    if (variant == 2) return true;
    const int64_t t256 = (int64_t)((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (t256 < 192) return false;                      // does not fill the 256 CUs
    // padded work of both tilings
    const int64_t t128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
    return (double)t256 * 4.0 / worth <= (double)t128;
}

template <int EPI, class ALoad>
int launch(const Problem& p, const ALoad& al, const Epilogue& e, hipStream_t st, const char* what, int batch = 1) {
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), (unsigned)batch), block(NTHR);
    hipLaunchKernelGGL((gemm128_kernel<EPI, ALoad>), grid, block, 0, st, p, al, e);
    YUME_CHECK_LAUNCH(what);
    return YUME_OK;
}

}  // namespace gemm_core
