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
    int64_t ldc, ldw, ldo, ldadd;
    int Tin, H, W, C, To, cout, kt, pt;
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
#define HN_DMA(voff, sbase, lds) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory")

#ifndef HN_ABLATE
#define HN_ABLATE 0       // experiments only (timing, wrong results): 1 = no DMA inside the loop, 2 = no waits / barriers, 4 = fragments read at tap 0 of a step only
#endif
template <int N>
__device__ __forceinline__ void wait_bar() {
    if constexpr (HN_ABLATE & 2) return;       // everything but the last N vector-memory operations of this wave has landed; all LDS reads are back
    asm volatile("s_waitcnt vmcnt(%c0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int NJ, int MI, int TH, int TW, int NW>
struct Geo {
    static_assert(NW == 4 || NW == 8, "one or two waves per SIMD");
    static_assert(TW % 16 == 0 && (TH * TW) / 16 == NW * MI, "NW waves x MI m-tiles of 16 columns cover the TH x TW region");
    static constexpr int MTR = TW / 16;                    // m-tiles per row
    static constexpr int HR = TH + 2, HC = TW + 2, NPOS = HR * HC;
    static constexpr int NPIECE = (NPOS + 15) / 16;        // pieces of 16 positions x 64 B
    static constexpr int PPW = (NPIECE + NW - 1) / NW;     // halo pieces per wave and step
    static constexpr int HALO_BYTES = PPW * NW * 1024;
    static constexpr int HPT = (PPW + 7) / 8;              // halo pieces issued per tap (taps 0..7 of the step before)
    static constexpr int NSW = (NJ + NW - 1) / NW;         // weight pieces (16 rows x 64 B = one n-tile) per wave and tap
    static constexpr int WSLAB = NSW * NW * 1024;
    static constexpr int LDS = 2 * HALO_BYTES + 3 * WSLAB;
    static constexpr int nh(int tap) {                     // halo pieces issued inside tap `tap` (0..8; taps -1 / -2 = taps 8 / 7 of the step before)
        const int t = tap < 0 ? tap + 9 : tap;
        int n = 0;
        for (int j = t * HPT; j < (t + 1) * HPT; ++j) n += (t < 8 && j < PPW) ? 1 : 0;
        return n;
    }
};

// per-thread state of the K walk (a struct handed to the tap template below: the tap index has to be a compile-time constant for the
// counted waits, and clang does not capture locals for asm operands inside generic lambdas)
template <int NJ, int MI, int TH, int TW, int NW>
struct State {
    using G = Geo<NJ, MI, TH, TW, NW>;
    const char* w;                 // weight base (bytes)
    char* smem;
    unsigned hoff[G::PPW];         // per-lane source offsets of this wave's halo pieces (0xffffffff: outside the image)
    unsigned woff[G::NSW];         // ... of its weight pieces
    unsigned lds_h, lds_w;         // LDS byte addresses of this wave's 1 KiB piece slot in halo buffer 0 / weight slab 0
    unsigned wfrag;                // weight fragment offset inside an n-tile's 1 KiB
    int pbase[MI];                 // halo position of m-tile i under tap (0, 0)
    int C, nslice, nstep, ntap, l4;
    f32x4 acc[MI][NJ];
#if HN_ABLATE & 4
    bf16x8_t wf_keep[NJ], xf_keep[MI];
#endif
};

// the weights of tap `tap` of step (dt, cs) into slab (global tap index) % 3 = tap % 3 (a step has 9 taps)
template <int NJ, int MI, int TH, int TW, int NW>
__device__ __forceinline__ void stage_w(State<NJ, MI, TH, TW, NW>& s, int dt, int cs, int tap) {
    using G = Geo<NJ, MI, TH, TW, NW>;
    const char* src = s.w + ((int64_t)(dt * 9 + tap) * s.C + cs * 32) * 2;
    const unsigned dst = s.lds_w + (unsigned)(tap % 3) * G::WSLAB;
#pragma unroll
    for (int q = 0; q < G::NSW; ++q) HN_DMA(s.woff[q], src, dst + q * (NW * 1024));
}

template <int TAP, int NJ, int MI, int TH, int TW, int NW>
__device__ __forceinline__ void tap_body(State<NJ, MI, TH, TW, NW>& s, int step, bool more, int dt, int cs, int dtn, int csn, const i32x4& srdn,
                                         const char* hb) {
    using G = Geo<NJ, MI, TH, TW, NW>;
    const int g = step * 9 + TAP;
    const unsigned soffn = (unsigned)csn * 64u;
    // slab g has landed and is visible, every wave is done with slab g - 1 (and, at tap 0, with the other halo buffer).
    // Issued behind slab g: [halo(tap - 2)] slab g + 1 [halo(tap - 1)] — counted exactly where all of them exist, else more is awaited
    if (g == 0) wait_bar<G::NSW>();                                  // prologue: halo 0 and slab 0 (slab 1 may stay in flight)
    else if (more && step > 0) wait_bar<G::nh(TAP - 2) + G::NSW + G::nh(TAP - 1)>();
    else if (more && TAP >= 2) wait_bar<G::nh(TAP - 2) + G::NSW + G::nh(TAP - 1)>();   // first step: no halo pieces were issued before its tap 0
    else if (more) wait_bar<G::NSW + G::nh(0)>();                    // first step, tap 1
    else if (g + 1 < s.ntap) wait_bar<G::NSW>();
    else wait_bar<0>();
    if constexpr (!(HN_ABLATE & 1)) {
        if constexpr (TAP + 2 < 9) stage_w(s, dt, cs, TAP + 2);
        else if (more) stage_w(s, dtn, csn, TAP + 2 - 9);
    }
    if (more && !(HN_ABLATE & 1)) {
#pragma unroll
        for (int j = TAP * G::HPT; j < (TAP + 1) * G::HPT; ++j)
            if (TAP < 8 && j < G::PPW) HN_DMAB(s.hoff[j], srdn, soffn, s.lds_h + (unsigned)((step + 1) & 1) * G::HALO_BYTES + j * (NW * 1024));
    }
    constexpr int dh = TAP / 3, dw = TAP % 3;
    const char* wb = s.smem + 2 * G::HALO_BYTES + (TAP % 3) * G::WSLAB + s.wfrag;
#if HN_ABLATE & 4
    bf16x8_t (&wf)[NJ] = s.wf_keep;
    bf16x8_t (&xf)[MI] = s.xf_keep;
    if constexpr (TAP == 0) {
#else
    bf16x8_t wf[NJ], xf[MI];
    {
#endif
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(wb + j * 1024);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int pos = s.pbase[i] + dh * G::HC + dw;
            xf[i] = *reinterpret_cast<const bf16x8_t*>(hb + (unsigned)pos * 64u + (unsigned)((s.l4 ^ ((pos >> 1) & 2)) << 4));
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) s.acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], s.acc[i][j], 0, 0, 0);
}
template <int TAP, int NJ, int MI, int TH, int TW, int NW>
__device__ __forceinline__ void run_taps(State<NJ, MI, TH, TW, NW>& s, int step, bool more, int dt, int cs, int dtn, int csn, const i32x4& srdn,
                                         const char* hb) {
    if constexpr (TAP < 9) {
        tap_body<TAP>(s, step, more, dt, cs, dtn, csn, srdn, hb);
        run_taps<TAP + 1>(s, step, more, dt, cs, dtn, csn, srdn, hb);
    }
}

template <int NJ, int MI, int TH, int TW, int NW, bool ADD>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv_halo_n_kernel(Params p) {
    using G = Geo<NJ, MI, TH, TW, NW>;
    __shared__ __attribute__((aligned(16))) char smem[G::LDS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    // tile: XCD x walks its own contiguous chunk of the (frame, row band, column band) order
    int start, count;
    const int ntile = p.To * p.tiles_h * p.tiles_w;
    xcd_chunk(ntile, blockIdx.x & 7, start, count);
    const int tile = start + (blockIdx.x >> 3);
    const int tw = tile % p.tiles_w, th = (tile / p.tiles_w) % p.tiles_h, to = tile / (p.tiles_w * p.tiles_h);
    const int h0 = th * TH, w0 = tw * TW;
    const unsigned ldc2 = (unsigned)p.ldc * 2u;

    State<NJ, MI, TH, TW, NW> s;
    s.w = reinterpret_cast<const char*>(p.w);
    s.smem = smem;
    s.C = p.C;
    s.l4 = l4;
    // ---- halo pieces of this wave: piece j covers halo positions 16 (4 j + wave) + (lane >> 2); slot lane & 3 holds chunk slot ^ ((pos >> 1) & 2)
#pragma unroll
    for (int j = 0; j < G::PPW; ++j) {
        const int pos = 16 * (NW * j + wave) + (lane >> 2);
        const int r = pos / G::HC, c = pos - r * G::HC;
        const int hi = h0 - 1 + r, wi = w0 - 1 + c;
        const bool ok = pos < G::NPOS && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
        s.hoff[j] = ok ? (unsigned)(hi * p.W + wi) * ldc2 + (unsigned)(((lane & 3) ^ ((pos >> 1) & 2)) << 4) : 0xffffffffu;
    }
    // ---- weight pieces of this wave: piece q = n-tile 4 q + wave (rows beyond cout read row cout - 1: their accumulator rows are never stored)
#pragma unroll
    for (int q = 0; q < G::NSW; ++q) {
        const int row = 16 * (NW * q + wave) + (lane >> 2);
        s.woff[q] = (unsigned)min(row, p.cout - 1) * (unsigned)p.ldw * 2u + (unsigned)(((lane & 3) ^ ((row >> 1) & 2)) << 4);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    s.lds_h = lds0 + wave * 1024;
    s.lds_w = lds0 + 2 * G::HALO_BYTES + wave * 1024;
    s.nslice = p.C / 32;
    s.nstep = p.kt * s.nslice;
    s.ntap = s.nstep * 9;
    // ---- fragment addresses. Weights: n-tile j, row l15, chunk l4 of the slab. Halo: m-tile i of this wave = row (wave MI + i) / MTR,
    //      columns 16 ((wave MI + i) % MTR) + l15; under tap (dh, dw) it reads halo position (row + dh) HC + column + dw
    s.wfrag = (unsigned)(l15 * 64 + ((l4 ^ ((l15 >> 1) & 2)) << 4));
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int mt = wave * MI + i;
        s.pbase[i] = (mt / G::MTR) * G::HC + (mt % G::MTR) * 16 + l15;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) s.acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: halo of step 0, weights of taps 0 and 1
    {
        const i32x4 srd0 = frame_srd(p, to - p.pt);
#pragma unroll
        for (int j = 0; j < G::PPW; ++j) HN_DMAB(s.hoff[j], srd0, 0u, s.lds_h + j * (NW * 1024));
        stage_w(s, 0, 0, 0);
        stage_w(s, 0, 0, 1);
    }
    int dt = 0, cs = 0;
    for (int step = 0; step < s.nstep; ++step) {
        const bool more = step + 1 < s.nstep;                            // (uniform) a next halo to stage
        int dtn = dt, csn = cs + 1;
        if (csn == s.nslice) { csn = 0; ++dtn; }
        const i32x4 srdn = frame_srd(p, to - p.pt + dtn);
        run_taps<0>(s, step, more, dt, cs, dtn, csn, srdn, smem + (step & 1) * G::HALO_BYTES);
        dt = dtn;
        cs = csn;
    }

    // ---- epilogue: a lane holds channels 16 j + 4 l4 .. + 3 of position (row, column 16 m + l15) of its MI m-tiles
    const int nch = 4 * l4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int mt = wave * MI + i;
        const int ho = h0 + mt / G::MTR, wo = w0 + (mt % G::MTR) * 16 + l15;
        if (ho >= p.H || wo >= p.W) continue;
        const int64_t pos = ((int64_t)to * p.H + ho) * p.W + wo;
        unsigned short* o = p.out + pos * p.ldo + nch;
        const unsigned short* a = ADD ? p.add + pos * p.ldadd + nch : nullptr;
        u32x2 a2[NJ];
        if (ADD) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) a2[j] = (16 * j + nch < p.cout) ? *reinterpret_cast<const u32x2*>(a + 16 * j) : u32x2{0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (16 * j + nch >= p.cout) continue;                 // whole groups of 4 channels (cout % 4 == 0)
            f32x4 v = s.acc[i][j];
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
}

// the shapes the kernel takes (host). The rule looks at the LAYER (channels, kernel, frame size), never at the number of frames in the
// launch: the grouped passes of a decoder and its one-latent-per-pass walk must take the same kernel for a layer (equal bits,
// tests/test_live_fullsize_gpu.py).
inline int instance(int64_t Cin, int64_t Cout) {
    if (Cin % 32 != 0 || Cin % 64 == 0) return 0;            // whole 64-wide K tiles are conv_w4's / the fast loader's
    if (Cout == 96) return 96;
    if (Cout == 160) return 160;
    return 0;
}
inline bool applies(int64_t Cin, int64_t Cout, int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int ups, int64_t Hin, int64_t Win,
                    int64_t Ho, int64_t Wo, int64_t ldc, int64_t ldo, int64_t ldw, int epi, int64_t ldadd) {
    static const bool on = [] { const char* v = getenv("YUME_CONV_HALO_N"); return !v || atoi(v) != 0; }();
    if (!on || ups || st != 1 || sh != 1 || sw != 1 || kh != 3 || kw != 3 || ph != 1 || pw != 1) return false;
    if (!((kt == 3 && pt == 2) || (kt == 1 && pt == 0))) return false;
    if (epi != YUME_EPI_BF16 && epi != YUME_CONV_EPI_ADD) return false;
    if (instance(Cin, Cout) == 0 || Ho != Hin || Wo != Win) return false;
    if ((ldc % 8) != 0 || (ldw % 8) != 0 || (ldo % 4) != 0 || ldo < Cout || (epi == YUME_CONV_EPI_ADD && ((ldadd % 4) != 0 || ldadd < Cout))) return false;
    if (Hin * Win * ldc * 2 >= 0x7fffff00ll || (Cout - 1) * ldw * 2 + 64 >= 0x7fffff00ll) return false;
    return Ho * Wo >= 16 * 1024;            // >= 32 (96 channels) / 64 (160) tiles per frame; smaller frames stay on the GEMM kernels
}

template <int NJ, int MI, int TH, int TW, int NW>
inline int launch_inst(Params hp, int64_t To, int64_t Ho, int64_t Wo, bool add, hipStream_t s) {
    hp.tiles_w = (int)((Wo + TW - 1) / TW);
    hp.tiles_h = (int)((Ho + TH - 1) / TH);
    const int64_t nt = To * hp.tiles_h * hp.tiles_w;
    if (nt >= (1ll << 31)) return -1;
    if (add) hipLaunchKernelGGL((conv_halo_n_kernel<NJ, MI, TH, TW, NW, true>), dim3((unsigned)nt), dim3(NW * 64), 0, s, hp);
    else hipLaunchKernelGGL((conv_halo_n_kernel<NJ, MI, TH, TW, NW, false>), dim3((unsigned)nt), dim3(NW * 64), 0, s, hp);
    return 0;
}

}  // namespace conv_halo_n
