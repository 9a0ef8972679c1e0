// conv_in.hpp — the FIRST convolution of the two VAE encoders: causal 3x3x3, stride 1, from the 3 (Wan2.1) / 12 (Wan2.2, patchified) input
// channels — 8 / 16 with the channel padding of the channels-last image — to the 96 / 160 channels of the first level
// (reference wan/modules/vae.py:291 `self.conv1 = CausalConv3d(3, dims[0], 3, padding=1)`, wan23/modules/vae2_2.py:525 likewise with 12). (r6)
//
// K = 27 taps x 8 (16) channels = 216 (432): on the gathering GEMM loader every 16-byte chunk of the A operand is one tap of one position with
// its own address arithmetic, and the launch is bound by neither the matrix pipe nor the 3.2 GB of output it writes — 3.06 ms for the 16.7 M
// positions of a Wan2.1 49-frame encode against 0.55 ms of HBM time (profiles/r6_vae21_encode_rocprofv3_kernel_stats_v2.csv).
// Here:
//   * a workgroup (4 waves) owns TH x TW output positions of one frame and all NJ x 16 output channels of the launch, and is persistent over
//     the tiles of its XCD's share;
//   * the WEIGHTS ARE RESIDENT IN REGISTERS for the whole launch: K = 7 (14) k-steps of 32, lane (n = lane & 15, kb = lane >> 4) holds
//     W[16 j + n][32 ks + 8 kb .. + 7] for all j, ks — 168 (280) registers; one wave per SIMD;
//   * the (TH + 2) x (TW + 2) x 3-frame halo of a tile is staged in LDS by plain 16-byte loads (out-of-image positions and a missing causal
//     cache = zeros): 16 (32) bytes per position, 31 (62) KiB;
//   * a k-step of the A operand is 4 taps x 8 channels (2 taps x 16): lane (position lane & 15, kb) reads the 16-byte record of "its" tap at the
//     shifted position — the tap offsets are per-lane constants; the taps 27.. of the last k-step read a zero record (their weights are
//     zero too, but 0 x NaN-bits would poison the sum);
//   * operand order as conv_halo_n: weights first, so a lane holds 4 consecutive channels of one position: 8-byte bf16 stores.
// The work is 42 (2 x 70) MFMAs per 16 positions: the kernel is bound by its output stream. Compiler-scheduled; nothing names registers.
// Roofline: HBM (2 * positions * Cout bytes written); MFMA work 2 * positions * Cout * 27 * Cin flop.
#pragma once
#include "gemm_core.hpp"

namespace conv_in {
using namespace gemm_core;

struct Params {
    const unsigned short* x;       // [Tin, H, W, ldc], ldc == CIN
    const unsigned short* cache;   // [2, H, W, ldc] or nullptr
    const unsigned short* w;       // [cout, ldw]: row n, column ((dt*3 + dh)*3 + dw) * CIN + c, zero padded to ldw
    const float* bias;             // [cout] or nullptr
    unsigned short* out;           // [To, H, W, ldo]
    int64_t ldw, ldo;
    int Tin, H, W, To, cout;
    int tiles_w, tiles_h;
};

template <int CIN, int NJ, int TH, int TW>
__global__ __launch_bounds__(256, 1) void conv_in_kernel(Params p) {
    constexpr int NKS = (27 * CIN + 31) / 32;            // 7 / 14 k-steps
    constexpr int REC = CIN * 2;                         // bytes per position
    constexpr int HH = TH + 2, HW = TW + 2;
    constexpr int FRAME = HH * HW * REC;
    constexpr int ZERO = 3 * FRAME;                      // a zero record behind the three frames
    constexpr int MT = TH * TW / 16;                     // m-tiles of 16 consecutive columns
    constexpr int MPW = MT / 4;                          // per wave
    constexpr int MI = 2;
    static_assert(TW % 16 == 0 && MT % 4 == 0 && MPW % MI == 0, "tile shape");
    static_assert(CIN == 8 || CIN == 16, "8 or 16 input channels");
    __shared__ __attribute__((aligned(16))) char halo[3 * FRAME + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kb = lane >> 4;

    // ---- weights and bias into registers (once per launch) ----
    bf16x8_t wf[NJ][NKS];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = 16 * j + l15;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (n < p.cout) v = *reinterpret_cast<const u32x4*>(p.w + (int64_t)n * p.ldw + 32 * ks + 8 * kb);
            wf[j][ks] = __builtin_bit_cast(bf16x8_t, v);
        }
    }
    f32x4 bias[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        bias[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && 16 * j + 4 * kb < p.cout) bias[j] = *reinterpret_cast<const f32x4*>(p.bias + 16 * j + 4 * kb);
    }
    // ---- per-lane byte offset of "its" tap in every k-step, relative to the record of the output position's halo origin ----
    int toff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int tap = CIN == 8 ? 4 * ks + kb : 2 * ks + (kb >> 1);
        const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
        toff[ks] = tap < 27 ? dt * FRAME + (dh * HW + dw) * REC + (CIN == 16 ? (kb & 1) * 16 : 0) : -1;
    }
    if (tid < 4) reinterpret_cast<u32x4*>(halo + ZERO)[tid] = u32x4{0u, 0u, 0u, 0u};

    const int ntile = p.To * p.tiles_h * p.tiles_w;
    int start, count;
    xcd_chunk(ntile, blockIdx.x & 7, start, count);
    const int per = gridDim.x >> 3;
    // A tile's halo (frames to-2 .. to, rows h0-1 .. h0+TH, columns w0-1 .. w0+TW) travels global -> registers -> LDS: the loads of tile k + 1
    // are issued before tile k is computed and stored into the (single) LDS image behind it, so their latency hides under the MFMAs.
    constexpr int NREC = (3 * HH * HW + 255) / 256;
    u32x4 hv0[NREC], hv1[CIN == 16 ? NREC : 1];
    auto tile_origin3 = [&](int k, int& to, int& h0, int& w0) {
        const int t = start + k;
        to = t / (p.tiles_w * p.tiles_h);
        const int rem = t - to * (p.tiles_w * p.tiles_h);
        h0 = (rem / p.tiles_w) * TH;
        w0 = (rem % p.tiles_w) * TW;
    };
    auto halo_load = [&](int k) {
        int to, h0, w0;
        tile_origin3(k, to, h0, w0);
#pragma unroll
        for (int u = 0; u < NREC; ++u) {
            const int r = tid + 256 * u;
            const int f = r / (HH * HW), q = r - f * (HH * HW), hr = q / HW, wc = q - hr * HW;
            const int ti = to + f - 2, hi = h0 + hr - 1, wi = w0 + wc - 1;
            const bool inside = r < 3 * HH * HW && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W && (ti >= 0 ? ti < p.Tin : p.cache != nullptr);
            const unsigned short* src = (ti >= 0 ? p.x + (int64_t)ti * p.H * p.W * CIN : p.cache + (int64_t)(ti + 2) * p.H * p.W * CIN) +
                                        ((int64_t)hi * p.W + wi) * CIN;
            hv0[u] = u32x4{0u, 0u, 0u, 0u};
            if (CIN == 16) hv1[u] = u32x4{0u, 0u, 0u, 0u};
            if (inside) {
                hv0[u] = *reinterpret_cast<const u32x4*>(src);
                if (CIN == 16) hv1[u] = *reinterpret_cast<const u32x4*>(src + 8);
            }
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int u = 0; u < NREC; ++u) {
            const int r = tid + 256 * u;
            if (r < 3 * HH * HW) {
                *reinterpret_cast<u32x4*>(halo + r * REC) = hv0[u];
                if (CIN == 16) *reinterpret_cast<u32x4*>(halo + r * REC + 16) = hv1[u];
            }
        }
    };
    int k = blockIdx.x >> 3;
    if (k < count) halo_load(k);
    for (; k < count; k += per) {
        int to, h0, w0;
        tile_origin3(k, to, h0, w0);
        __syncthreads();                                   // the previous tile's fragment reads are over
        halo_store();
        __syncthreads();
        if (k + per < count) halo_load(k + per);           // in flight while this tile computes
        // ---- this wave's m-tiles, MI at a time ----
#pragma unroll 1
        for (int g = 0; g < MPW / MI; ++g) {
            f32x4 acc[MI][NJ];
            int base[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int mt = wave * MPW + g * MI + i;
                base[i] = ((mt / (TW / 16)) * HW + (mt % (TW / 16)) * 16 + l15) * REC;
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                bf16x8_t a[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(halo + (toff[ks] >= 0 ? base[i] + toff[ks] : ZERO));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], a[i], acc[i][j], 0, 0, 0);
            }
            // ---- epilogue: a lane holds channels 16 j + 4 kb .. + 3 of position (row, column) of each of its m-tiles ----
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int mt = wave * MPW + g * MI + i;
                const int ho = h0 + mt / (TW / 16), wo = w0 + (mt % (TW / 16)) * 16 + l15;
                if (ho >= p.H || wo >= p.W) continue;
                unsigned short* o = p.out + (((int64_t)to * p.H + ho) * p.W + wo) * p.ldo + 4 * kb;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (16 * j + 4 * kb >= p.cout) continue;
                    const f32x4 v = acc[i][j] + bias[j];
                    u32x2 ov;
                    ov[0] = pack_bf16x2(v[0], v[1]);
                    ov[1] = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(o + 16 * j) = ov;
                }
            }
        }
    }
}

// the layers the kernel takes (host): the encoders' first convolution — 8 -> 96 (one launch) and 16 -> 160 (two launches of 80 channels)
inline bool applies(int64_t Cin, int64_t Cout, int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int ups, int64_t Hin, int64_t Win,
                    int64_t Ho, int64_t Wo, int64_t ldc, int64_t ldo, int64_t ldw, int epi) {
    static const bool on = [] { const char* v = getenv("YUME_CONV_IN"); return !v || atoi(v) != 0; }();
    if (!on || ups || epi != YUME_EPI_BF16) return false;
    if (kt != 3 || kh != 3 || kw != 3 || st != 1 || sh != 1 || sw != 1 || pt != 2 || ph != 1 || pw != 1) return false;
    if (!((Cin == 8 && Cout == 96) || (Cin == 16 && Cout == 160)) || ldc != Cin) return false;
    if (Ho != Hin || Wo != Win || (ldo % 4) != 0 || ldo < Cout || (ldw % 8) != 0 || ldw < 32 * ((27 * Cin + 31) / 32)) return false;
    if (Hin * Win * Cin * 2 >= 0x7fffff00ll) return false;
    return Ho * Wo >= 16 * 1024;
}

template <int CIN, int NJ>
inline int launch(Params hp, int64_t To, int64_t Ho, int64_t Wo, hipStream_t s) {
    constexpr int TH = 8, TW = 64;
    hp.tiles_w = (int)((Wo + TW - 1) / TW);
    hp.tiles_h = (int)((Ho + TH - 1) / TH);
    const int64_t nt = To * hp.tiles_h * hp.tiles_w;
    if (nt >= (1ll << 31)) return -1;
    static const int ncu = [] {
        int d = 0, n = 256;
        if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess) n = 256;
        return n > 0 ? n : 256;
    }();
    const int64_t per_xcd = (nt + 7) / 8 < (ncu + 7) / 8 ? (nt + 7) / 8 : (ncu + 7) / 8;
    hipLaunchKernelGGL((conv_in_kernel<CIN, NJ, TH, TW>), dim3((unsigned)(8 * per_xcd)), dim3(256), 0, s, hp);
    return 0;
}

}  // namespace conv_in
