// conv_halo.hpp — causal 3x3x3 convolution with FEW output channels (Cout <= 16: the decoder's head conv 256 -> 12 at full resolution,
// reference wan23/modules/vae2_2.py:737 `self.head = Sequential(RMS_norm, SiLU, CausalConv3d(dims[-1], 12, 3, padding=1))`).
//
// As an implicit GEMM with a 256-wide (or 128-wide) N tile the head wastes 94 % of its MFMAs and, worse, moves its A operand 27 times
// (one 256 x 64 slice per tap and channel tile) through L2 -> LDS: 340 GB per chunk decode, ~35 ms of the 398 ms (r3 kernel trace). Here
// a workgroup owns 4 rows x 64 columns of one output frame (256 positions x 16 channels) and, per input frame dt and 64-channel slice,
// stages the 6 x 66 HALO of that region in LDS once (LDS-DMA, 8 positions x 128 B per piece, chunk-swizzled on the source side, positions
// outside the image = out-of-range lanes of the frame's buffer descriptor = zeros); the 9 in-plane taps are fragment reads out of the
// halo at shifted positions. Input bytes through L2 -> LDS: 1.55 x 3 frames instead of 27 x. The weights of a step (9 taps x 64
// channels x 16 rows = 18 KiB) are MFMA B fragments in registers, loaded one step ahead.
//   wave w = output row h0 + w: 4 tiles of 16 positions x 16 channels (4 x f32x4 accumulators), 72 MFMAs per step;
//   halo buffers: 2 x 52 KiB (step s+1 lands while step s computes); one wait + barrier per step.
// HBM / L2 bound by design: algorithmic bytes = input read once per dt (3 x) + output written once.
#pragma once
#include "gemm_core.hpp"

namespace conv_halo {
using namespace gemm_core;

constexpr int TH = 4, TW = 64;                 // output tile: rows x columns
constexpr int HR = TH + 2, HC = TW + 2;        // halo
constexpr int NPOS = HR * HC;                  // 396 positions
constexpr int NPIECE = (NPOS + 7) / 8;         // 50 pieces of 8 positions
constexpr int PPW = (NPIECE + 3) / 4;          // 13 pieces per wave
constexpr int HALO_BYTES = PPW * 4 * 1024;     // 53,248 B per buffer (the tail pieces hold no position)

struct Params {
    const unsigned short* x;       // [Tin, H, W, ldc]
    const unsigned short* cache;   // [2, H, W, ldc] or nullptr
    const unsigned short* w;       // [16, ldw]: row n, column ((dt*3 + dh)*3 + dw) * C + c
    const float* bias;             // [>= 16] or nullptr
    unsigned short* out;           // [To, H, W, ldo]
    int64_t ldc, ldw, ldo;
    int Tin, H, W, C, To, cout;
    int tiles_w, tiles_h;
};

#define HALO_DMAB(voff, srd, soff, lds) \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(lds) : "memory")

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

template <int UNUSED = 0>
__global__ __launch_bounds__(256, 1) void conv_halo16_kernel(Params p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * HALO_BYTES];
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

    // ---- DMA pieces of this wave: piece j covers halo positions 8 (4 j + wave) + (lane >> 3), 16-byte chunk (lane & 7) ^ (position & 7)
    unsigned voff[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pos = 8 * (4 * j + wave) + (lane >> 3);
        const int r = pos / HC, c = pos - r * HC;
        const int hi = h0 - 1 + r, wi = w0 - 1 + c;
        const bool ok = pos < NPOS && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
        voff[j] = ok ? (unsigned)(hi * p.W + wi) * ldc2 + (unsigned)(((lane & 7) ^ (pos & 7)) << 4) : 0xffffffffu;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;
    auto stage = [&](int buf, i32x4 srd, unsigned soff) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) HALO_DMAB(voff[j], srd, soff, lds0 + buf * HALO_BYTES + j * 4096);
    };

    // ---- fragment addresses: output position (row wave, column 16 m + l15) under tap (dh, dw) reads halo position
    //      (wave + dh) * HC + 16 m + l15 + dw; its 16-byte chunk 4 ks + l4 sits at chunk ^ (position & 7)
    const int pbase = wave * HC + l15;                          // + dh * HC + dw (+ 16 m: a multiple of 8, leaves position & 7)

    // ---- weights of a step: B fragments [tap][ks]: W[n = l15][((dt*9 + tap) * C + c0 + 32 ks + 8 l4 ...)]
    //      (rows n >= cout do not exist in the weight: their lanes read row cout - 1 and are zeroed)
    const unsigned short* wrow = p.w + (int64_t)min(l15, p.cout - 1) * p.ldw + l4 * 8;
    const bool wreal = l15 < p.cout;
    const int nc = p.C / 64, nstep = 3 * nc;
    auto load_b = [&](bf16x8_t (&b)[9][2], int step) {
        const int dt = step / nc, c0 = (step - dt * nc) * 64;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 v = *reinterpret_cast<const u32x4*>(wrow + (int64_t)(dt * 9 + tap) * p.C + c0 + ks * 32);
                if (!wreal) v = u32x4{0u, 0u, 0u, 0u};
                b[tap][ks] = __builtin_bit_cast(bf16x8_t, v);
            }
    };

    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8_t bcur[9][2], bnext[9][2];
    stage(0, frame_srd(p, to - 2), 0u);
    load_b(bcur, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");

    for (int step = 0; step < nstep; ++step) {
        const int buf = step & 1;
        if (step + 1 < nstep) {                                          // (uniform) next halo into the other buffer, next weights into registers
            const int dtn = (step + 1) / nc, cn = (step + 1) - dtn * nc;
            stage(buf ^ 1, frame_srd(p, to - 2 + dtn), (unsigned)cn * 128u);
            load_b(bnext, step + 1);
        }
        const char* hb = smem + buf * HALO_BYTES;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int pos = pbase + dh * HC + dw;
                const unsigned a0 = (unsigned)pos * 128u + (unsigned)((l4 ^ (pos & 7)) << 4);       // k-step 0; k-step 1: chunk 4 + l4 = the same ^ 64 bytes
                const unsigned a1 = a0 ^ 64u;
                bf16x8_t f0[4], f1[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f0[m] = *reinterpret_cast<const bf16x8_t*>(hb + a0 + m * 2048);
                    f1[m] = *reinterpret_cast<const bf16x8_t*>(hb + a1 + m * 2048);
                }
                // (the four accumulators alternate: a dependent MFMA comes back 4 MFMAs later)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bcur[dh * 3 + dw][0], f0[m], acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bcur[dh * 3 + dw][1], f1[m], acc[m], 0, 0, 0);
            }
        // every wave is done reading this buffer, the next halo has landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            bcur[tap][0] = bnext[tap][0];
            bcur[tap][1] = bnext[tap][1];
        }
    }

    // ---- epilogue: a lane holds channels 4 l4 .. 4 l4 + 3 of position (row h0 + wave, column w0 + 16 m + l15)
    f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (4 * l4 + q < p.cout) b4[q] = p.bias[4 * l4 + q];
    }
    const int ho = h0 + wave;
    if (ho < p.H && 4 * l4 < p.cout) {            // whole groups of 4 channels (cout % 4 == 0): the row padding [cout, ldo) of `out` is not touched
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int wo = w0 + 16 * m + l15;
            if (wo >= p.W) continue;
            const f32x4 v = acc[m] + b4;
            u32x2 o;
            o[0] = pack_bf16x2(v[0], v[1]);
            o[1] = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<u32x2*>(p.out + (((int64_t)to * p.H + ho) * p.W + wo) * p.ldo + 4 * l4) = o;
        }
    }
}

// the shapes the kernel takes (host)
inline bool applies(int64_t Cin, int64_t Cout, int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int ups, int64_t Hin, int64_t Win,
                    int64_t Ho, int64_t Wo, int64_t ldc, int64_t ldo, int64_t ldw, int epi) {
    static const bool on = [] { const char* v = getenv("YUME_CONV_HALO"); return !v || atoi(v) != 0; }();
    if (!on || epi != YUME_EPI_BF16 || ups || st != 1 || sh != 1 || sw != 1) return false;
    if (kt != 3 || kh != 3 || kw != 3 || pt != 2 || ph != 1 || pw != 1 || Cout > 16 || (Cin % 64) != 0) return false;
    if (Ho != Hin || Wo != Win || (ldo != 8 && ldo != 16) || ldo < Cout || (Cout % 4) != 0 || (ldc % 8) != 0 || (ldw % 8) != 0) return false;
    if (Hin * Win * ldc * 2 >= 0x7fffff00ll) return false;
    return Ho * Wo >= 64 * 1024;           // the full-resolution head; small frames stay on the GEMM kernels
}

}  // namespace conv_halo
