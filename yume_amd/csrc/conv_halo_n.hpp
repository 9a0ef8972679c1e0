// conv_halo_n.hpp — stride-1 3x3 (x3) convolutions whose channel counts are whole 32-channel slices but NOT whole 64-wide K tiles or
// 128 / 256-wide N tiles: the 96-channel full-resolution level of the Wan2.1 VAE (reference wan/modules/vae.py:369-472 decoder,
// :255-366 encoder, dim = 96) and the 160-channel level of the Wan2.2 encoder (wan23/modules/vae2_2.py:506-622, dim = 160). (r6)
//
// Until r5 these ran on the generic-loader 128 x 128 GEMM kernel at 0.44 PFLOP/s and were half of a Wan2.1 decode
// (profiles/r5_vae21_conv_selection.txt): Cin = 96 is not a whole number of 64-wide K tiles, so every 16-byte chunk of the A operand
// computed its own tap address in VALU code; an N tile of 128 wastes 25 % of the MFMAs; and the implicit GEMM moves every input position
// 27 times through L2 -> LDS. As an implicit GEMM with an N tile of 96 the kernel would be L2 -> LDS bound whatever its schedule (96 flop
// per A byte against the 128 of the 256 x 256 tile, which already needs 46 GB/s per CU). So this is conv_halo.hpp's structure instead,
// generalised from "16 output channels, weights in registers" to an N tile of NJ x 16 channels with the weights staged through LDS:
//   * a workgroup (4 waves, one per SIMD) owns TH x TW output positions of ONE output frame and ALL NJ x 16 output channels;
//     wave w owns MI of the 4 MI m-tiles (16 consecutive columns of one row each): MI x NJ accumulator tiles of 16 x 16;
//   * K is walked (dt, 32-channel slice cs, dh, dw). Per step (dt, cs) the (TH + 2) x (TW + 2) HALO of the region is staged in LDS once
//     (LDS-DMA, 16 positions x 64 B per piece; positions outside the image = out-of-range lanes of the frame's buffer descriptor = zeros;
//     a missing causal cache = a descriptor of zero records) and the 9 in-plane taps are fragment reads at shifted positions: input bytes
//     through L2 -> LDS 1.3 x 3 frames instead of 27 x. The halo of step s + 1 lands while step s computes (two buffers);
//   * the weights of a tap — NJ x 16 rows x 64 B — go through a ring of three LDS slabs, two taps ahead of their use;
//   * 64-byte rows (positions / weight rows): 16-byte chunk c of row r sits at slot c ^ ((r >> 1) & 2) — conflict-free ds_read_b128 fragment
//     reads of 16 consecutive rows at ANY alignment of the first row (the taps shift the halo position by 0 / 1 / 2);
//   * one counted vmcnt + lgkmcnt(0) + barrier per tap (MI x NJ MFMAs = 768 / 640 matrix-pipe clocks for the 96 / 160-channel instance);
//   * MFMA operand order (weights first): a lane holds 4 consecutive channels of one position — 8-byte bf16 stores into the channels-last
//     output, the shortcut of a residual block (EPI ADD) read the same way.
// The compiler schedules the tap body (as in conv_halo.hpp); nothing here names registers.
// Roofline: MFMA bf16 dense; algorithmic work 2 * positions * Cout * Cin * kt * 9 flop per launch.
#pragma once
#include "gemm_core.hpp"

namespace conv_halo_n {
using namespace gemm_core;

struct Params {
    const unsigned short* x;       // [Tin, H, W, ldc]
    const unsigned short* cache;   // [2, H, W, ldc] or nullptr
    const unsigned short* w;       // [cout, ldw]: row n, column ((dt*3 + dh)*3 + dw) * C + c
    const float* bias;             // [cout] or nullptr
    unsigned short* out;           // [To, H, W, ldo]
    const unsigned short* add;     // EPI ADD: [To, H, W, ldadd] or nullptr
    const float* gamma;            // EPI RMS_SILU: [cout] — out = SiLU(RMS_norm(acc + bias) * gamma), the norm over the cout channels of a position
    int64_t ldc, ldw, ldo, ldadd;
    int Tin, H, W, C, To, cout, kt, pt;      // H, W: the INPUT frame (the output frame is 2H x 2W with the folded nearest-2x upsample)
    int tiles_w, tiles_h;
};

__device__ __forceinline__ i32x4 frame_srd(const Params& p, int ti) {
    const int64_t frame = (int64_t)p.H * p.W * p.ldc;
    const unsigned short* base = ti >= 0 ? p.x + (int64_t)ti * frame : p.cache + (int64_t)(ti + 2) * frame;
    const bool have = (ti >= 0 && ti < p.Tin) || (ti < 0 && ti >= -2 && p.cache != nullptr);
    const uint64_t b = (uint64_t)(uintptr_t)base;
    i32x4 d;
    d[0] = (int)(unsigned)(b & 0xffffffffu);
    d[1] = (int)(unsigned)((b >> 32) & 0xffffu);
    d[2] = have ? 0x7fffffff : 0;
    d[3] = 0x00020000;
    return d;
}

#define HN_DMAB(voff, srd, soff, lds) \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(lds) : "memory")

#ifndef HN_ABLATE
#define HN_ABLATE 0       // experiments only (timing, wrong results): 1 = no DMA inside the loop, 2 = no waits / barriers, 4 = no fragment reads behind a tile's first tap
#endif
template <int N>
__device__ __forceinline__ void wait_bar() {       // everything but the last N vector-memory operations of this wave has landed; all LDS reads are back
    if constexpr (HN_ABLATE & 2) return;
    asm volatile("s_waitcnt vmcnt(%c0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int NJ, int MI, int TH, int TW, int NW, bool UPS = false>
struct Geo {
    static_assert(NW == 4 || NW == 8, "one or two waves per SIMD");
    static_assert(TW % 16 == 0 && (TH * TW) / 16 == NW * MI, "NW waves x MI m-tiles of 16 columns cover the TH x TW region");
    static_assert(!UPS || (TH % 2 == 0 && TW % 2 == 0), "an upsampled tile starts on an even row / column");
    static constexpr int MTR = TW / 16;                    // m-tiles per row
    // UPS: the convolution reads the nearest-2x upsampled view of its input (nn.Upsample folded in, wan/modules/vae.py:114-128): the halo is
    // the INPUT region (TH/2 + 2) x (TW/2 + 2) under the TH x TW output tile, output (2a+e, 2b+f) under tap (dh, dw) reads input row
    // (2a + e + dh - 1) >> 1, column (2b + f + dw - 1) >> 1
    static constexpr int HR = (UPS ? TH / 2 : TH) + 2, HC = (UPS ? TW / 2 : TW) + 2, NPOS = HR * HC;
    static constexpr int NPIECE = (NPOS + 15) / 16;        // pieces of 16 positions x 64 B
    static constexpr int PPW = (NPIECE + NW - 1) / NW;     // halo pieces per wave and step
    static constexpr int HPT = (PPW + 7) / 8;              // halo pieces issued per tap (taps 0..7 of the step before; the surplus slots are dummies)
    static constexpr int HALO_BYTES = (PPW + 1) * NW * 1024;   // (+ one piece slot per wave where the dummy pieces' zeros land)
    static constexpr int NSW = (NJ + NW - 1) / NW;         // weight pieces (16 rows x 64 B = one n-tile) per wave and tap
    static constexpr int WSLAB = NSW * NW * 1024;
    static constexpr int LDS = 2 * HALO_BYTES + 3 * WSLAB;
    // EVERY tap issues exactly NSW weight pieces and HPT halo pieces, real or dummy (all lanes out of the descriptor's range: nothing is
    // fetched, zeros land in LDS where nobody reads). Behind the slab a tap waits for come [halo of tap - 2] slab + 1 [halo of tap - 1]:
    static constexpr int KWAIT = 2 * HPT + NSW;
};

// (r6, third build) The tap loop is a RUN-TIME loop: the first two builds unrolled the 9 taps of a step (compile-time wait counts) and the
// compiler, pipelining across them, ran out of registers (44-110 VGPRs to scratch in the persistent form — and scratch traffic counts in vmcnt,
// which the counted waits below cannot tolerate). With uniform issue counts per tap one wait immediate serves every tap.
// EPI: 0 bias only, 1 + the shortcut (ADD), 2 RMS_norm over the output channels + SiLU (the residual block's second norm, fused: r6)
template <int NJ, int MI, int TH, int TW, int NW, int EPI, bool UPS = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv_halo_n_kernel(Params p) {
    using G = Geo<NJ, MI, TH, TW, NW, UPS>;
    __shared__ __attribute__((aligned(16))) char smem[G::LDS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    // tiles: XCD x walks its own contiguous chunk of the (frame, row band, column band) order, its g workgroups interleaved in it. PERSISTENT:
    // a tile is ~40 us of MFMAs between a prologue that waits for its first halo and an epilogue of 8-byte stores, and with one workgroup per
    // CU nothing overlapped them (with every DMA, wait and fragment read removed the one-tile-per-workgroup build ran at 1.5 PF,
    // profiles/r6_conv_halo_n_ablations.log). During the last step of a tile the workgroup stages the first halo and the first two weight slabs
    // of its NEXT tile.
    int start, count;
    const int ntile = p.To * p.tiles_h * p.tiles_w;
    xcd_chunk(ntile, blockIdx.x & 7, start, count);
    const int gwg = (int)(gridDim.x >> 3), idx = (int)(blockIdx.x >> 3);
    if (idx >= count) return;                                        // (workgroup-uniform)
    int tile = start + idx;
    const int tend = start + count;
    auto coords = [&](int t, int& to, int& h0, int& w0) {
        const int tw = t % p.tiles_w, th = (t / p.tiles_w) % p.tiles_h;
        to = t / (p.tiles_w * p.tiles_h);
        h0 = th * TH;
        w0 = tw * TW;
    };
    int to, h0, w0;
    coords(tile, to, h0, w0);
    const unsigned ldc2 = (unsigned)p.ldc * 2u;
    const int nslice = p.C / 32, nstep = p.kt * nslice;

    // source offset of this lane in halo piece j (positions 16 (NW j + wave) + (lane >> 2)) of the tile at (hh, ww); slot lane & 3 holds chunk
    // slot ^ ((pos >> 1) & 2); outside the image / beyond the halo: out of range (zeros)
    auto halo_off = [&](int j, int hh, int ww) -> unsigned {
        const int pos = 16 * (NW * j + wave) + (lane >> 2);
        const int r = pos / G::HC, c = pos - r * G::HC;
        const int hi = (UPS ? hh / 2 : hh) - 1 + r, wi = (UPS ? ww / 2 : ww) - 1 + c;
        const bool ok = pos < G::NPOS && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
        return ok ? (unsigned)(hi * p.W + wi) * ldc2 + (unsigned)(((lane & 3) ^ ((pos >> 1) & 2)) << 4) : 0xffffffffu;
    };
    // weight pieces of this wave: piece q = n-tile NW q + wave; n-tiles that do not exist are dummies
    unsigned woff[G::NSW];
#pragma unroll
    for (int q = 0; q < G::NSW; ++q) {
        const int row = 16 * (NW * q + wave) + (lane >> 2);
        woff[q] = row < p.cout ? (unsigned)row * (unsigned)p.ldw * 2u + (unsigned)(((lane & 3) ^ ((row >> 1) & 2)) << 4) : 0xffffffffu;
    }
    i32x4 wsrd;                                                         // the weight array as a raw buffer (rows 0 .. cout - 1)
    {
        const uint64_t b = (uint64_t)(uintptr_t)p.w;
        wsrd[0] = (int)(unsigned)(b & 0xffffffffu);
        wsrd[1] = (int)(unsigned)((b >> 32) & 0xffffu);
        wsrd[2] = 0x7fffffff;
        wsrd[3] = 0x00020000;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lds_h = lds0 + wave * 1024, lds_w = lds0 + 2 * G::HALO_BYTES + wave * 1024;
    // fragment addresses. Weights: n-tile j, row l15, chunk l4 of the slab. Halo: m-tile i of this wave = row (wave MI + i) / MTR,
    // columns 16 ((wave MI + i) % MTR) + l15; under tap (dh, dw) it reads halo position (row + dh) HC + column + dw
    const unsigned wfrag = (unsigned)(l15 * 64 + ((l4 ^ ((l15 >> 1) & 2)) << 4));
    int pbase[MI];                                                      // (UPS: the m-tile's output column of this lane, minus 1)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int mt = wave * MI + i;
        pbase[i] = UPS ? (mt % G::MTR) * 16 + l15 - 1 : (mt / G::MTR) * G::HC + (mt % G::MTR) * 16 + l15;
    }
    // weights of (dt, cs, tap): byte offset inside a weight row; slab of a tap = tap % 3
    auto w_soff = [&](int dt, int cs, int tap) -> unsigned { return (unsigned)(((dt * 9 + tap) * p.C + cs * 32) * 2); };
    auto stage_w = [&](unsigned soff, int slot, bool real) {
#pragma unroll
        for (int q = 0; q < G::NSW; ++q) HN_DMAB(real ? woff[q] : 0xffffffffu, wsrd, soff, lds_w + (unsigned)slot * G::WSLAB + q * (NW * 1024));
    };

    // ---- prologue: halo of step 0 of the first tile (all 8 HPT piece slots), weights of taps 0 and 1
    {
        const i32x4 srd0 = frame_srd(p, to - p.pt);
#pragma unroll
        for (int j = 0; j < G::PPW; ++j) HN_DMAB(halo_off(j, h0, w0), srd0, 0u, lds_h + j * (NW * 1024));
        stage_w(w_soff(0, 0, 0), 0, true);
        stage_w(w_soff(0, 0, 1), 1, true);
    }
    unsigned gs = 0;                                                     // steps done by this workgroup: the halo buffer of a step is gs & 1
    const int nch = 4 * l4;
    f32x4 acc[MI][NJ];
    for (;;) {
        const int tnext = tile + gwg;
        const bool has_next = tnext < tend;                              // (uniform)
        int ton = to, h0n = h0, w0n = w0;
        if (has_next) coords(tnext, ton, h0n, w0n);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int dt = 0, cs = 0;
        for (int step = 0; step < nstep; ++step, ++gs) {
            const bool last = step + 1 == nstep;
            const bool more = !last || has_next;                         // (uniform) a next halo to stage: the next step's, or the next tile's first
            int dtn = dt, csn = cs + 1;
            if (csn == nslice) { csn = 0; ++dtn; }
            if (last) { dtn = 0; csn = 0; }
            const int hh = last ? h0n : h0, ww = last ? w0n : w0;        // whose halo is staged during this step
            const i32x4 srdn = frame_srd(p, (last ? ton : to) - p.pt + dtn);
            const unsigned soffn = (unsigned)csn * 64u;
            const char* hb = smem + (gs & 1u) * G::HALO_BYTES;
            const unsigned nbuf = lds_h + ((gs + 1u) & 1u) * G::HALO_BYTES;
            int toff = 0, dh = 0, dw = 0;                                // halo offset of the tap: dh HC + dw
            // (written as a loop, unrolled by the compiler: the piece indices, slab slots and tap offsets become literals — the rolled form
            // hipcc chose for one instance on its own measured 842 against 1145 TFLOP/s — while nothing is scheduled across the taps' barriers)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // slab `tap` has landed and is visible, every wave is done with the slab before it (and, at tap 0, with the other halo buffer).
                // The first tap of a tile waits for everything: its halo and first slabs were issued in front of the previous tile's epilogue
                // stores, and loads and stores do not retire in order with each other — only vmcnt(0) is safe behind stores.
                if (step == 0 && tap == 0) wait_bar<0>();
                else wait_bar<G::KWAIT>();
                if (!(HN_ABLATE & 1)) {
                    // weights two taps ahead (this step's, the next step's, or the next tile's first two), then this tap's share of the next halo
                    const int t2 = tap + 2;
                    if (t2 < 9) stage_w(w_soff(dt, cs, t2), t2 % 3, true);
                    else stage_w(w_soff(dtn, csn, t2 - 9), (t2 - 9) % 3, more);
#pragma unroll
                    for (int k = 0; k < G::HPT; ++k) {
                        const int j = tap * G::HPT + k;
                        // the LDS slot depends on the loop counters only (a piece index the step has -> its slot, else the spare slot); whether
                        // anything is FETCHED also depends on `more` — kept out of the slot's select, which the compiler otherwise folds into the
                        // per-lane select of the source offset and hands the asm's M0 operand a VGPR
                        const bool exists = tap < 8 && j < G::PPW;
                        const unsigned slot = nbuf + (unsigned)(exists ? j : G::PPW) * (NW * 1024);
                        HN_DMAB((more && exists) ? halo_off(j, hh, ww) : 0xffffffffu, srdn, soffn, slot);
                    }
                }
                const char* wb = smem + 2 * G::HALO_BYTES + (tap % 3) * G::WSLAB + wfrag;
                bf16x8_t wf[NJ], xf[MI];
#pragma unroll
                for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(wb + j * 1024);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    int pos = pbase[i] + toff;
                    if constexpr (UPS) pos = ((((wave * MI + i) / G::MTR + dh - 1) >> 1) + 1) * G::HC + ((pbase[i] + dw) >> 1) + 1;
                    xf[i] = *reinterpret_cast<const bf16x8_t*>(hb + (unsigned)pos * 64u + (unsigned)((l4 ^ ((pos >> 1) & 2)) << 4));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
                toff += (tap % 3 == 2) ? G::HC - 2 : 1;
                if (++dw == 3) { dw = 0; ++dh; }
            }
            dt = dtn;
            cs = csn;
        }

        // ---- epilogue: a lane holds channels 16 j + 4 l4 .. + 3 of position (row, column 16 m + l15) of its MI m-tiles
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int mt = wave * MI + i;
            const int ho = h0 + mt / G::MTR, wo = w0 + (mt % G::MTR) * 16 + l15;
            const int OH = UPS ? 2 * p.H : p.H, OW = UPS ? 2 * p.W : p.W;
            if (ho >= OH || wo >= OW) continue;
            const int64_t pos = ((int64_t)to * OH + ho) * OW + wo;
            unsigned short* o = p.out + pos * p.ldo + nch;
            if constexpr (EPI == 2) {
                // The position's NJ * 16 channels sit in this lane (4 per n-tile) and in the three lanes 16, 32, 48 further on: RMS_norm of
                // vae.py's RMS_norm (F.normalize over the channels * sqrt(C) * gamma, eps 1e-12 on the norm) on the fp32 accumulators, then SiLU
                // — what the stand-alone kernel (vae_ops.hip) does to the bf16 image of the same values one launch later.
                // (two passes over the accumulators instead of a copy of the row: the 96-channel instance has 256 registers per lane)
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 v = acc[i][j];
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + 16 * j + nch);
                    ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                const float inv = sqrtf((float)(NJ * 16)) / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 v = acc[i][j];
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + 16 * j + nch);
                    const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + 16 * j + nch);
                    float r[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t = v[k] * inv * g[k];
                        r[k] = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
                    }
                    u32x2 ov;
                    ov[0] = pack_bf16x2(r[0], r[1]);
                    ov[1] = pack_bf16x2(r[2], r[3]);
                    *reinterpret_cast<u32x2*>(o + 16 * j) = ov;
                }
                continue;
            }
            constexpr bool ADD = EPI == 1;
            const unsigned short* a = ADD ? p.add + pos * p.ldadd + nch : nullptr;
            u32x2 a2[NJ];
            if (ADD) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) a2[j] = (16 * j + nch < p.cout) ? *reinterpret_cast<const u32x2*>(a + 16 * j) : u32x2{0u, 0u};
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (16 * j + nch >= p.cout) continue;                 // whole groups of 4 channels (cout % 4 == 0)
                f32x4 v = acc[i][j];
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + 16 * j + nch);
                if (ADD) {
                    v[0] += bf16_to_f32((unsigned short)(a2[j][0] & 0xffffu));
                    v[1] += bf16_to_f32((unsigned short)(a2[j][0] >> 16));
                    v[2] += bf16_to_f32((unsigned short)(a2[j][1] & 0xffffu));
                    v[3] += bf16_to_f32((unsigned short)(a2[j][1] >> 16));
                }
                u32x2 ov;
                ov[0] = pack_bf16x2(v[0], v[1]);
                ov[1] = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<u32x2*>(o + 16 * j) = ov;
            }
        }
        if (!has_next) break;
        tile = tnext;
        to = ton;
        h0 = h0n;
        w0 = w0n;
    }
}

// the shapes the kernel takes (host). The rule looks at the LAYER (channels, kernel, frame size), never at the number of frames in the
// launch: the grouped passes of a decoder and its one-latent-per-pass walk must take the same kernel for a layer (equal bits,
// tests/test_live_fullsize_gpu.py).
inline int instance(int64_t Cin, int64_t Cout, int ups = 0) {
    if (ups) return (Cin % 32 == 0 && Cout == 96) ? 96 : 0;  // the 192 -> 96 upsample convolution of the Wan2.1 decoder (37 % of a 256-wide N tile)
    if (Cin % 32 != 0 || Cin % 64 == 0) return 0;            // whole 64-wide K tiles are conv_w4's / the fast loader's
    if (Cout == 96 || Cout == 192) return 96;                // (192 = two launches over 96 output channels each: the level's widening convolution)
    if (Cout == 160 || Cout == 320) return 160;              // (320 = two launches of 160)
    if (Cout <= 16) return 16;                               // the Wan2.1 decoder's head, 96 -> 3 (+1) channels at full resolution (conv_halo.hpp's case at Cin % 64 != 0)
    return 0;
}
inline bool applies(int64_t Cin, int64_t Cout, int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int ups, int64_t Hin, int64_t Win,
                    int64_t Ho, int64_t Wo, int64_t ldc, int64_t ldo, int64_t ldw, int epi, int64_t ldadd) {
    static const bool on = [] { const char* v = getenv("YUME_CONV_HALO_N"); return !v || atoi(v) != 0; }();
    if (!on || st != 1 || sh != 1 || sw != 1 || kh != 3 || kw != 3 || ph != 1 || pw != 1) return false;
    if (!((kt == 3 && pt == 2) || (kt == 1 && pt == 0))) return false;
    if (epi != YUME_EPI_BF16 && epi != YUME_CONV_EPI_ADD && epi != YUME_CONV_EPI_RMS_SILU) return false;
    if (epi == YUME_CONV_EPI_RMS_SILU && (ups || (Cout != 96 && Cout != 160))) return false;      // the whole channel row in one workgroup
    if (instance(Cin, Cout, ups) == 0) return false;
    if (ups ? (Ho != 2 * Hin || Wo != 2 * Win || kt != 1) : (Ho != Hin || Wo != Win)) return false;
    if ((ldc % 8) != 0 || (ldw % 8) != 0 || (ldo % 4) != 0 || ldo < Cout || (Cout % 4) != 0 || (epi == YUME_CONV_EPI_ADD && ((ldadd % 4) != 0 || ldadd < Cout))) return false;
    if (Hin * Win * ldc * 2 >= 0x7fffff00ll || (Cout - 1) * ldw * 2 + 64 >= 0x7fffff00ll) return false;
    return Ho * Wo >= 16 * 1024;            // >= 32 (96 channels) / 64 (160) tiles per frame; smaller frames stay on the GEMM kernels
}

template <int NJ, int MI, int TH, int TW, int NW, bool UPS = false>
inline int launch_inst(Params hp, int64_t To, int64_t Ho, int64_t Wo, int epi_kind, hipStream_t s) {
    hp.tiles_w = (int)((Wo + TW - 1) / TW);
    hp.tiles_h = (int)((Ho + TH - 1) / TH);
    const int64_t nt = To * hp.tiles_h * hp.tiles_w;
    if (nt >= (1ll << 31)) return -1;
    // persistent: one workgroup per CU (its LDS admits no second), 32 per XCD, each walking tiles idx, idx + 32, ... of its XCD's chunk
    // (YUME_CONV_HALO_PERSIST=0: one workgroup per tile, the first build, for A/B)
    static const bool persist = [] { const char* v = getenv("YUME_CONV_HALO_PERSIST"); return !v || atoi(v) != 0; }();
    static const int ncu = [] {
        int d = 0, n = 256;
        if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess) n = 256;
        return n > 0 ? n : 256;
    }();
    const int64_t per_xcd = persist ? (ncu + 7) / 8 : (nt + 7) / 8;
    const unsigned grid = (unsigned)(8 * ((nt + 7) / 8 < per_xcd ? (nt + 7) / 8 : per_xcd));
    if constexpr (!UPS && NJ > 1) {
        if (epi_kind == 2) {
            hipLaunchKernelGGL((conv_halo_n_kernel<NJ, MI, TH, TW, NW, 2, UPS>), dim3(grid), dim3(NW * 64), 0, s, hp);
            return 0;
        }
    }
    if (epi_kind == 2) return -3;
    if (epi_kind == 1) hipLaunchKernelGGL((conv_halo_n_kernel<NJ, MI, TH, TW, NW, 1, UPS>), dim3(grid), dim3(NW * 64), 0, s, hp);
    else hipLaunchKernelGGL((conv_halo_n_kernel<NJ, MI, TH, TW, NW, 0, UPS>), dim3(grid), dim3(NW * 64), 0, s, hp);
    return 0;
}

}  // namespace conv_halo_n
