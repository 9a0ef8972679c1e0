// vae_ops.hip — the HBM-bound pieces of the causal 3D VAE on channels-last bf16 activations:
// RMS_norm(+SiLU), DupUp3D / AvgDown3D shortcut adds, row softmax of the single-head attention, and the
// NCTHW fp32 <-> channels-last bf16 layout conversions (with patchify / unpatchify / latent scaling / clamp).
// Roofline: HBM. Algorithmic bytes per element: rmsnorm_silu 2 read + 2 write.
#include "common.hpp"

namespace {

// ---- RMS_norm (+SiLU): LPR lanes cooperate on one row, 64/LPR rows per wave, 4 waves per workgroup ----
template <int LPR, int NV>
__global__ __launch_bounds__(256) void rmsnorm_silu_kernel(const unsigned short* x, int64_t ldx, int64_t M,
                                                           int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int silu_on,
                                                           unsigned short* y, int64_t ldy) {      // (x == y allowed: a row is read whole before it is written)
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row < M;
    const int nvec = C >> 3;
    u16x8 v[NV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = sub + i * LPR;
        if (live && vi < nvec) {
            v[i] = *reinterpret_cast<const u16x8*>(x + row * ldx + 8 * vi);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = bf16_to_f32(v[i][j]);
                ss += f * f;
            }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = sub + i * LPR;
        if (live && vi < nvec) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + 8 * vi);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(gamma + 8 * vi + 4);
            f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
            if (beta) {
                b0 = *reinterpret_cast<const f32x4*>(beta + 8 * vi);
                b1 = *reinterpret_cast<const f32x4*>(beta + 8 * vi + 4);
            }
            float o[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = bf16_to_f32(v[i][j]) * inv * g0[j] + b0[j];
                o[4 + j] = bf16_to_f32(v[i][4 + j]) * inv * g1[j] + b1[j];
            }
            if (silu_on) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = silu(o[j]);
            }
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = pack_bf16x2(o[2 * j], o[2 * j + 1]);
            *reinterpret_cast<u32x4*>(y + row * ldy + 8 * vi) = w;
        }
    }
}

// ---- RMS_norm (+SiLU), every lane busy: G lanes (a power of two) own one row and NVL 16-byte vectors of it each (G * NVL = C / 8), 64 / G rows
// per wave, RI such row sets per wave. The kernel above is bound by VALU issue, not by HBM (3.4-4.0 TB/s at every VAE level with a quarter
// to three eighths of the lanes idle: 12 of 16 lanes at 96 channels, 20 of 32 at 160; the IEEE division of its SiLU is ~10 issue slots per
// element): here 96 / 192 / 384 channels run as G = 4 / 8 / 16 with NVL = 3 and 160 / 320 / 640 as NVL = 5, SiLU is x * rcp(1 + exp2(-x log2 e))
// (one v_exp_f32, one v_rcp_f32: 1 ulp, below the bf16 rounding of the result) and the arithmetic runs two elements per issue on the
// packed fp32 ALU.
__device__ __forceinline__ f32x2_t silu2_fast(f32x2_t x) {
    const f32x2_t k = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.0f, 1.0f};
    const f32x2_t u = x * k;
    const f32x2_t e = {__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};
    const f32x2_t d = e + one;
    const f32x2_t r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return x * r;
}

template <int G, int NVL, int RI>
__global__ __launch_bounds__(256) void rmsnorm_silu_g_kernel(const unsigned short* x, int64_t ldx, int64_t M, int C,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int silu_on, unsigned short* y, int64_t ldy) {      // (x == y allowed)
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G;
    const int64_t row0 = (int64_t)blockIdx.x * (RI * 4 * RPW) + (threadIdx.x >> 6) * RPW + lane / G;
    u32x4 v[RI][NVL];
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const int64_t row = row0 + (int64_t)r * (4 * RPW);
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            v[r][i] = u32x4{0u, 0u, 0u, 0u};
            if (row < M) v[r][i] = *reinterpret_cast<const u32x4*>(x + row * ldx + 8 * (sub + i * G));
        }
    }
    float inv[RI];
    const float sqrt_c = sqrtf((float)C);
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        f32x2_t s2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NVL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2_t f = {__uint_as_float(v[r][i][j] << 16), __uint_as_float(v[r][i][j] & 0xffff0000u)};
                s2 += f * f;
            }
        float ss = s2[0] + s2[1];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        inv[r] = sqrt_c / fmaxf(sqrtf(ss), 1e-12f);
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
        const int c0 = 8 * (sub + i * G);
        f32x2_t g[4], b[4];
        {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c0), g1 = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
            g[0] = f32x2_t{g0[0], g0[1]}; g[1] = f32x2_t{g0[2], g0[3]}; g[2] = f32x2_t{g1[0], g1[1]}; g[3] = f32x2_t{g1[2], g1[3]};
            f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
            if (beta) {
                b0 = *reinterpret_cast<const f32x4*>(beta + c0);
                b1 = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
            }
            b[0] = f32x2_t{b0[0], b0[1]}; b[1] = f32x2_t{b0[2], b0[3]}; b[2] = f32x2_t{b1[0], b1[1]}; b[3] = f32x2_t{b1[2], b1[3]};
        }
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const int64_t row = row0 + (int64_t)r * (4 * RPW);
            if (row >= M) continue;
            const f32x2_t iv = {inv[r], inv[r]};
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2_t f = {__uint_as_float(v[r][i][j] << 16), __uint_as_float(v[r][i][j] & 0xffff0000u)};
                f32x2_t o = (f * iv) * g[j] + b[j];
                if (silu_on) o = silu2_fast(o);
                w[j] = pack_bf16x2(o[0], o[1]);
            }
            *reinterpret_cast<u32x4*>(y + row * ldy + 8 * (sub + i * G)) = w;
        }
    }
}

// ---- RMS_norm (+SiLU), flat form for contiguous rows (ldx == ldy == C): a wave owns NVL * 64 consecutive 16-byte vectors = R = NVL * 64 / NVEC
// whole rows (NVEC = C / 8 vectors per row) and lane l takes vectors l, 64 + l, ... of that chunk, so every load and every store instruction
// of the wave is one contiguous KiB whatever the row length (the grouped form above reads a 96- or 160-channel row as 64-byte pieces,
// 16 rows apart per instruction: 4.0-4.4 TB/s against 5.7 at 192 / 384 channels). The sum of squares crosses lanes through the wave's own
// 4 * NVL * 64 bytes of LDS: every lane leaves the partial sum of each of its vectors, lane r < R adds row r's NVEC partials in index order and
// leaves the row's scale, every lane picks up the scale of the row of each of its vectors. gamma (and beta) belong to (lane, i) alone
// (a chunk starts on a row boundary), so they stay in registers across the RI chunks of a wave.
template <int NVEC, int NVL, int RI, bool SILU, bool BETA>
__global__ __launch_bounds__(256) void rmsnorm_silu_flat_kernel(const unsigned short* x, int64_t M, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, unsigned short* y) {      // (x == y allowed: a wave reads its chunk whole before it writes)
    constexpr int CH = NVL * 64;                 // vectors per chunk
    constexpr int R = CH / NVEC;                 // rows per chunk
    static_assert(R * NVEC == CH && R <= 64 && (NVEC % 4) == 0, "a chunk is whole rows");
    __shared__ __attribute__((aligned(16))) float part[4][RI][CH];
    __shared__ float scale[4][RI][R];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t total = M * NVEC;
    const int64_t vb0 = ((int64_t)blockIdx.x * 4 + wave) * RI * CH;
    u32x4 v[RI][NVL];
#pragma unroll
    for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            const int64_t f = vb0 + r * CH + i * 64 + lane;
            v[r][i] = u32x4{0u, 0u, 0u, 0u};
            if (f < total) v[r][i] = *reinterpret_cast<const u32x4*>(x + 8 * f);
        }
#pragma unroll
    for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            f32x2_t s2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2_t f = {__uint_as_float(v[r][i][j] << 16), __uint_as_float(v[r][i][j] & 0xffff0000u)};
                s2 += f * f;
            }
            part[wave][r][i * 64 + lane] = s2[0] + s2[1];
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float sqrt_c = sqrtf((float)(NVEC * 8));
    if (lane < R) {
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            float ss = 0.f;
#pragma unroll
            for (int k = 0; k < NVEC / 4; ++k) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(&part[wave][r][lane * NVEC + 4 * k]);
                ss += (q[0] + q[1]) + (q[2] + q[3]);
            }
            scale[wave][r][lane] = sqrt_c / fmaxf(sqrtf(ss), 1e-12f);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
        const int fl = i * 64 + lane;
        const int rowl = fl / NVEC, c0 = 8 * (fl % NVEC);
        f32x2_t g[4], b[4];
        {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c0), g1 = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
            g[0] = f32x2_t{g0[0], g0[1]}; g[1] = f32x2_t{g0[2], g0[3]}; g[2] = f32x2_t{g1[0], g1[1]}; g[3] = f32x2_t{g1[2], g1[3]};
            if (BETA) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c0), b1 = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
                b[0] = f32x2_t{b0[0], b0[1]}; b[1] = f32x2_t{b0[2], b0[3]}; b[2] = f32x2_t{b1[0], b1[1]}; b[3] = f32x2_t{b1[2], b1[3]};
            }
        }
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const int64_t f = vb0 + r * CH + fl;
            if (f >= total) continue;
            const float inv = scale[wave][r][rowl];
            const f32x2_t iv = {inv, inv};
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2_t e = {__uint_as_float(v[r][i][j] << 16), __uint_as_float(v[r][i][j] & 0xffff0000u)};
                f32x2_t o = (e * iv) * g[j];
                if (BETA) o += b[j];
                if (SILU) o = silu2_fast(o);
                w[j] = pack_bf16x2(o[0], o[1]);
            }
            *reinterpret_cast<u32x4*>(y + 8 * f) = w;
        }
    }
}

// ---- DupUp3D add: one thread per (output position, 8 output channels) ----
__global__ __launch_bounds__(256) void dupup_add_kernel(const unsigned short* __restrict__ x, int64_t ldx, int Tin, int Hin,
                                                        int Win, int Cin, unsigned short* __restrict__ y, int64_t ldy,
                                                        int To, int Cout, int ft, int fs, int toff) {
    const int Ho = Hin * fs, Wo = Win * fs;
    const int cv = Cout >> 3;
    const int64_t total = (int64_t)To * Ho * Wo * cv;
    const int F = ft * fs * fs;
    const int repeats = Cout * F / Cin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cv);
        const int64_t pos = i / cv;
        const int wo = (int)(pos % Wo);
        const int ho = (int)((pos / Wo) % Ho);
        const int to = (int)(pos / ((int64_t)Wo * Ho));
        const int tt = to + toff;
        const int ti = tt / ft, a = tt % ft;
        const int hi = ho / fs, b = ho % fs, wi = wo / fs, c = wo % fs;
        const int phase = (a * fs + b) * fs + c;
        const unsigned short* xr = x + (((int64_t)ti * Hin + hi) * Win + wi) * ldx;
        unsigned short* yr = y + pos * ldy + 8 * c8;
        u32x4 yv = *reinterpret_cast<const u32x4*>(yr);
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[2 * j] = bf16_to_f32((unsigned short)(yv[j] & 0xffffu));
            o[2 * j + 1] = bf16_to_f32((unsigned short)(yv[j] >> 16));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int oc = 8 * c8 + j;
            o[j] += bf16_to_f32(xr[(oc * F + phase) / repeats]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = pack_bf16x2(o[2 * j], o[2 * j + 1]);
        *reinterpret_cast<u32x4*>(yr) = yv;
    }
}

// ---- AvgDown3D add: one thread per (output position, output channel) ----
__global__ __launch_bounds__(256) void avgdown_add_kernel(const unsigned short* __restrict__ x, int64_t ldx, int Tin, int Hin,
                                                          int Win, int Cin, unsigned short* __restrict__ y, int64_t ldy,
                                                          int Cout, int ft, int fs) {
    const int padt = (ft - Tin % ft) % ft;
    const int To = (Tin + padt) / ft, Ho = Hin / fs, Wo = Win / fs;
    const int F = ft * fs * fs;
    const int G = Cin * F / Cout;
    const int64_t total = (int64_t)To * Ho * Wo * Cout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int oc = (int)(i % Cout);
        const int64_t pos = i / Cout;
        const int wo = (int)(pos % Wo);
        const int ho = (int)((pos / Wo) % Ho);
        const int to = (int)(pos / ((int64_t)Wo * Ho));
        float s = 0.f;
        for (int g = 0; g < G; ++g) {
            const int idx = oc * G + g;            // index into the (ci, a, b, c) unfolded channel axis
            const int ci = idx / F, ph = idx % F;
            const int c = ph % fs, b = (ph / fs) % fs, a = ph / (fs * fs);
            const int ti = to * ft + a - padt;
            if (ti >= 0) s += bf16_to_f32(x[(((int64_t)ti * Hin + ho * fs + b) * Win + wo * fs + c) * ldx + ci]);
        }
        unsigned short* yp = y + pos * ldy + oc;
        *yp = f32_to_bf16(bf16_to_f32(*yp) + s / (float)G);
    }
}

// ---- the two shortcut adds again, without the per-thread 64-bit div / mod chains and 2-byte gathers of the kernels above (those ran at
// 2.1 TB/s on the 6.5 GB read-modify-write of the Wan2.2 decoder's last DupUp3D: bound by integer VALU work, not by HBM) ----
// One grid row per output line (frame, image row): the frame / row / phase arithmetic is per workgroup; a thread owns 8 output channels of one
// output pixel and finds its pixel by one 32-bit division (a shift when the vector count per pixel is a power of two).
// DupUp3D where repeats | F (Cin >= Cout, Q = Cin / Cout in {1, 2, 4}): output channel oc reads input channel oc * Q + phase / repeats, so the 8
// output channels of a thread come out of Q consecutive 16-byte vectors of the input pixel.
template <int Q>
__global__ __launch_bounds__(256) void dupup_add_lines_kernel(const unsigned short* __restrict__ x, int64_t ldx, int Hin, int Win,
                                                              unsigned short* __restrict__ y, int64_t ldy, int Ho, int Wo, int cv, int cv_shift,
                                                              int ft, int fs, int toff, int rep_shift) {
    const int line = blockIdx.y;
    const int to = line / Ho, ho = line - to * Ho;
    const int tt = to + toff, ti = tt / ft, a = tt - ti * ft;
    const int hi = ho / fs, b = ho - hi * fs;
    const unsigned short* xl = x + ((int64_t)ti * Hin + hi) * Win * ldx;
    unsigned short* yl = y + (int64_t)line * Wo * ldy;
    const int n = Wo * cv;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int wo = cv_shift >= 0 ? (i >> cv_shift) : (i / cv);
        const int c8 = i - wo * cv;
        const int wi = wo / fs, c = wo - wi * fs;
        const int off = (((a * fs + b) * fs + c) >> rep_shift);           // < Q
        const unsigned short* xr = xl + (int64_t)wi * ldx + 8 * Q * c8;
        unsigned short* yr = yl + (int64_t)wo * ldy + 8 * c8;
        u32x4 xv[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) xv[k] = *reinterpret_cast<const u32x4*>(xr + 8 * k);
        u32x4 yv = *reinterpret_cast<const u32x4*>(yr);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // element j * Q + off of the Q vectors: dword (j * Q + off) >> 1, half (j * Q + off) & 1
            unsigned int w;
            if (Q == 1) w = xv[0][j >> 1];
            else if (Q == 2) w = xv[j >> 2][j & 3];
            else w = (off >> 1) ? xv[j >> 1][2 * (j & 1) + 1] : xv[j >> 1][2 * (j & 1)];
            const bool hi_half = Q == 1 ? (j & 1) != 0 : (off & 1) != 0;
            const float xf = __uint_as_float(hi_half ? (w & 0xffff0000u) : (w << 16));
            const unsigned int yw = yv[j >> 1];
            o[j] = __uint_as_float((j & 1) ? (yw & 0xffff0000u) : (yw << 16)) + xf;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = pack_bf16x2(o[2 * j], o[2 * j + 1]);
        *reinterpret_cast<u32x4*>(yr) = yv;
    }
}

// AvgDown3D where G | F (RR = F / G in {1, 2, 4, 8}): output channel oc averages the G phases (oc % RR) * G ... of input channel oc / RR, so the 8
// output channels of a thread read 8 / RR consecutive input channels at each of the F phase positions; the phases are added in the order
// of the kernel above (same bits).
template <int RR>
__global__ __launch_bounds__(256) void avgdown_add_lines_kernel(const unsigned short* __restrict__ x, int64_t ldx, int Hin, int Win,
                                                                unsigned short* __restrict__ y, int64_t ldy, int Ho, int Wo, int cv, int cv_shift,
                                                                int ft, int fs, int G, int padt) {
    const int line = blockIdx.y;
    const int to = line / Ho, ho = line - to * Ho;
    unsigned short* yl = y + (int64_t)line * Wo * ldy;
    const int n = Wo * cv, F = ft * fs * fs;
    const float inv_g = (float)G;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int wo = cv_shift >= 0 ? (i >> cv_shift) : (i / cv);
        const int c8 = i - wo * cv;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int p = 0; p < F; ++p) {
            const int c = p % fs, b = (p / fs) % fs, a = p / (fs * fs);
            const int ti = to * ft + a - padt;
            if (ti < 0) continue;
            const int cls = p / G;
            const unsigned short* xr = x + (((int64_t)ti * Hin + ho * fs + b) * Win + wo * fs + c) * ldx + (8 / RR) * c8;
            float e[8 / RR];
            if (RR == 1) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(xr);
#pragma unroll
                for (int j = 0; j < 8 / RR; ++j) e[j] = __uint_as_float((j & 1) ? (v[(j >> 1) & 3] & 0xffff0000u) : (v[(j >> 1) & 3] << 16));
            } else if (RR == 2) {
                const u32x2 v = *reinterpret_cast<const u32x2*>(xr);
#pragma unroll
                for (int j = 0; j < 8 / RR; ++j) e[j] = __uint_as_float((j & 1) ? (v[(j >> 1) & 1] & 0xffff0000u) : (v[(j >> 1) & 1] << 16));
            } else if (RR == 4) {
                const unsigned int v = *reinterpret_cast<const unsigned int*>(xr);
                e[0] = __uint_as_float(v << 16);
                e[(8 / RR) - 1] = __uint_as_float(v & 0xffff0000u);
            } else {
                e[0] = bf16_to_f32(*xr);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if ((j % RR) == cls) acc[j] += e[j / RR];
        }
        unsigned short* yr = yl + (int64_t)wo * ldy + 8 * c8;
        u32x4 yv = *reinterpret_cast<const u32x4*>(yr);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned int yw = yv[j >> 1];
            o[j] = __uint_as_float((j & 1) ? (yw & 0xffff0000u) : (yw << 16)) + acc[j] / inv_g;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = pack_bf16x2(o[2 * j], o[2 * j + 1]);
        *reinterpret_cast<u32x4*>(yr) = yv;
    }
}

// ---- row softmax: one workgroup per row, three passes over a (cache resident) fp32 row ----
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int64_t lds_, int n, float scale,
                                                           unsigned short* __restrict__ P, int64_t ldp) {
    __shared__ float red[8];
    const float* s = S + (int64_t)blockIdx.x * lds_;
    unsigned short* p = P + (int64_t)blockIdx.x * ldp;
    float mx = -3.0e38f;
    for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, s[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const float c = scale * 1.4426950408889634f;
    float sum = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) sum += __builtin_amdgcn_exp2f((s[i] - mx) * c);
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int i = threadIdx.x; i < (int)ldp; i += 256)
        p[i] = i < n ? f32_to_bf16(__builtin_amdgcn_exp2f((s[i] - mx) * c) * inv) : (unsigned short)0;
}

// ---- row softmax, the row read once: NV 16-byte vectors of the fp32 row per thread stay in registers between the maximum, the sum and the
// write (n <= 1024 * NV; lds_ a multiple of 4 so that the last vector of a row is inside its allocation, ldp a multiple of 4). The
// kernel above makes three scalar passes per row (178 us per 8160 x 8160 call of the Wan2.1 middle attention against 73 us of HBM time). ----
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const float* __restrict__ S, int64_t lds_, int n, float scale,
                                                               unsigned short* __restrict__ P, int64_t ldp) {
    __shared__ float red[8];
    const float* s = S + (int64_t)blockIdx.x * lds_;
    unsigned short* p = P + (int64_t)blockIdx.x * ldp;
    f32x4 v[NV];
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = 4 * (threadIdx.x + 256 * k);
        v[k] = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        if (i < n) {
            v[k] = *reinterpret_cast<const f32x4*>(s + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[k][j] = i + j < n ? v[k][j] : -3.0e38f;
        }
        mx = fmaxf(fmaxf(mx, fmaxf(v[k][0], v[k][1])), fmaxf(v[k][2], v[k][3]));
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const float c = scale * 1.4426950408889634f;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[k][j] = __builtin_amdgcn_exp2f((v[k][j] - mx) * c);      // masked entries: exp2 of a huge negative number = 0
            sum += v[k][j];
        }
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = 4 * (threadIdx.x + 256 * k);
        if (i < (int)ldp) {
            u32x2 w;
            w[0] = pack_bf16x2(v[k][0] * inv, v[k][1] * inv);
            w[1] = pack_bf16x2(v[k][2] * inv, v[k][3] * inv);
            *reinterpret_cast<u32x2*>(p + i) = w;
        }
    }
    for (int i = 4 * 256 * NV + threadIdx.x; i < (int)ldp; i += 256) p[i] = 0;       // padding beyond the register window
}

// ---- NCTHW -> channels-last (+ patchify, per-channel affine) ----
template <bool INBF16>
__global__ __launch_bounds__(256) void pack_input_kernel(const void* __restrict__ xv, int C, int T, int H, int W, int ps,
                                                         const float* __restrict__ mul, const float* __restrict__ add,
                                                         unsigned short* __restrict__ out, int Cpad) {
    const int Ho = H / ps, Wo = W / ps;
    const int64_t total = (int64_t)T * Ho * Wo * Cpad;
    const int Cv = C * ps * ps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int oc = (int)(i % Cpad);
        const int64_t pos = i / Cpad;
        unsigned short val = 0;
        if (oc < Cv) {
            const int wo = (int)(pos % Wo);
            const int ho = (int)((pos / Wo) % Ho);
            const int t = (int)(pos / ((int64_t)Wo * Ho));
            // oc = c*ps*ps + r*ps + q  ->  x[c, t, ho*ps + q, wo*ps + r]
            const int q = oc % ps, r = (oc / ps) % ps, c = oc / (ps * ps);
            const int64_t idx = (((int64_t)c * T + t) * H + ho * ps + q) * W + wo * ps + r;
            float f = INBF16 ? bf16_to_f32(reinterpret_cast<const unsigned short*>(xv)[idx])
                             : reinterpret_cast<const float*>(xv)[idx];
            if (mul) f = f * mul[c] + add[c];
            val = f32_to_bf16(f);
        }
        out[i] = val;
    }
}

// ---- channels-last -> NCTHW fp32 (+ unpatchify, (x - sub) * mul, clamp) ----
__global__ __launch_bounds__(256) void unpack_output_kernel(const unsigned short* __restrict__ x, int64_t ldx, int T, int H,
                                                            int W, int Cv, int ps, const float* __restrict__ sub,
                                                            const float* __restrict__ mul, float lo, float hi,
                                                            float* __restrict__ out) {
    const int Co = Cv / (ps * ps), Ho = H * ps, Wo = W * ps;
    const int64_t total = (int64_t)Co * T * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % Wo);
        const int h = (int)((i / Wo) % Ho);
        const int t = (int)((i / ((int64_t)Wo * Ho)) % T);
        const int c = (int)(i / ((int64_t)Wo * Ho * T));
        const int q = h % ps, r = w % ps;
        const int ic = c * ps * ps + r * ps + q;
        float f = bf16_to_f32(x[(((int64_t)t * H + h / ps) * W + w / ps) * ldx + ic]);
        if (mul) f = (f - sub[c]) * mul[c];
        if (lo < hi) f = fminf(fmaxf(f, lo), hi);
        out[i] = f;
    }
}

// fp32 [C, T, H, W] in [-1, 1] -> uint8 [T, H, W, C]: one thread = 4 consecutive w of all C channels
// (C coalesced 16-byte loads, one contiguous 4*C-byte store). HBM-bound: 4 B read + 1 B written per element.
// TRUNC: the web app's own conversion (webapp_single_gpu.py:117-121): ((x.clamp(-1, 1) + 1) / 2 * 255).byte() — truncation, no rounding
template <int C, bool TRUNC>
__global__ __launch_bounds__(256) void frames_u8_kernel(const float* __restrict__ x, int64_t plane /* T*H*W */, int64_t nquad,
                                                        unsigned char* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nquad; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned char b[4 * C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + c * plane + 4 * i);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // diffusers VideoProcessor: denormalize (x * 0.5 + 0.5).clamp(0, 1), then (x * 255).round() -> uint8 (half to even)
                if (TRUNC) {
                    const float f = __fmul_rn(__fadd_rn(fminf(fmaxf(v[q], -1.f), 1.f), 1.f), 0.5f);
                    b[q * C + c] = (unsigned char)(int)__fmul_rn(f, 255.f);        // (int): toward zero, like Tensor.byte()
                } else {
                    float f = __fadd_rn(__fmul_rn(v[q], 0.5f), 0.5f);
                    f = fminf(fmaxf(f, 0.f), 1.f);
                    b[q * C + c] = (unsigned char)(int)rintf(__fmul_rn(f, 255.f));
                }
            }
        }
        unsigned char* dst = out + 4 * C * i;
        if (C == 3 || C == 1 || C == 4 || C == 2) {
#pragma unroll
            for (int w = 0; w < C; ++w) {
                unsigned int u = (unsigned)b[4 * w] | ((unsigned)b[4 * w + 1] << 8) | ((unsigned)b[4 * w + 2] << 16) | ((unsigned)b[4 * w + 3] << 24);
                reinterpret_cast<unsigned int*>(dst)[w] = u;
            }
        }
    }
}

// T5 attention rows (wan/modules/t5.py:95-112): P[h,i,:n] = softmax_j(S[h,i,j] + bias[h, j - i + n - 1]), no scaling;
// one wave per row, n <= 1024; P[h,i,n:ldp] = 0 (the K padding of the following P.V product)
__global__ __launch_bounds__(256) void softmax_bias_kernel(const float* __restrict__ S, int64_t lds, int64_t strideS, int n,
                                                           const float* __restrict__ bias, int64_t ldb,
                                                           unsigned short* __restrict__ P, int64_t ldp, int64_t strideP,
                                                           int64_t rows_total) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_total) return;
    const int64_t h = row / n;
    const int i = (int)(row - h * n);
    const float* s = S + h * strideS + (int64_t)i * lds;
    const float* b = bias + h * ldb + (n - 1 - i);
    unsigned short* o = P + h * strideP + (int64_t)i * ldp;
    float v[16];
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = lane + 64 * k;
        v[k] = j < n ? s[j] + b[j] : -3.0e38f;
        mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = lane + 64 * k;
        v[k] = j < n ? __expf(v[k] - mx) : 0.f;
        sum += v[k];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = lane + 64 * k;
        if (j < ldp) o[j] = f32_to_bf16(v[k] * inv);
    }
}

inline unsigned grid_for(int64_t total, int per_block = 256, int64_t cap = 16384) {
    int64_t nb = (total + per_block - 1) / per_block;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

}  // namespace

extern "C" int yume_vae_rmsnorm_silu(const void* x, int64_t ldx, int64_t M, int64_t C, const float* gamma, const float* beta,
                                     int silu_on, void* y, int64_t ldy, void* stream) {
    YUME_REQUIRE(x && gamma && y, "vae_rmsnorm_silu: NULL pointer");
    YUME_REQUIRE(M > 0 && C > 0 && (C % 8) == 0 && C <= 4096 && (ldx % 8) == 0 && (ldy % 8) == 0, "vae_rmsnorm_silu: C=%lld must be a multiple of 8 and <= 4096", (long long)C);
    hipStream_t st = (hipStream_t)stream;
    const unsigned short* xp = (const unsigned short*)x;
    unsigned short* yp = (unsigned short*)y;
    const int nvec = (int)(C / 8);
#define LAUNCH_RMS(LPR, NV)                                                                                          \
    hipLaunchKernelGGL((rmsnorm_silu_kernel<LPR, NV>), dim3((unsigned)((M + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)))), \
                       dim3(256), 0, st, xp, ldx, M, (int)C, gamma, beta, silu_on, yp, ldy)
    // the every-lane-busy form where C / 8 = G * NVL with G a power of two <= 64 and NVL in {1, 3, 5} (YUME_VAE_NORM_G=0: the form above)
    static const bool grouped = [] { const char* v = getenv("YUME_VAE_NORM_G"); return !v || atoi(v) != 0; }();
    int G = 1;
    while (G < 64 && (nvec % (2 * G)) == 0) G *= 2;
    const int nvl = nvec / G;
#define LAUNCH_RMS_G(GG, NVL, RI)                                                                                                    \
    hipLaunchKernelGGL((rmsnorm_silu_g_kernel<GG, NVL, RI>), dim3((unsigned)((M + RI * 4 * (64 / GG) - 1) / (RI * 4 * (64 / GG)))), \
                       dim3(256), 0, st, xp, ldx, M, (int)C, gamma, beta, silu_on, yp, ldy)
#define LAUNCH_RMS_GN(NVL, RI)                                                                              \
    switch (G) {                                                                                            \
        case 1: LAUNCH_RMS_G(1, NVL, RI); break;   case 2: LAUNCH_RMS_G(2, NVL, RI); break;                \
        case 4: LAUNCH_RMS_G(4, NVL, RI); break;   case 8: LAUNCH_RMS_G(8, NVL, RI); break;                \
        case 16: LAUNCH_RMS_G(16, NVL, RI); break; case 32: LAUNCH_RMS_G(32, NVL, RI); break;              \
        default: LAUNCH_RMS_G(64, NVL, RI); break;                                                          \
    }
    // contiguous rows of 96 / 192 / 384 or 160 / 320 / 640 channels: the flat form (YUME_VAE_NORM_FLAT=0: the grouped one)
    static const bool flat = [] { const char* v = getenv("YUME_VAE_NORM_FLAT"); return !v || atoi(v) != 0; }();
#define LAUNCH_RMS_F(NVEC, NVL, RI)                                                                                                         \
    do {                                                                                                                                    \
        const unsigned nb = (unsigned)((M * NVEC + (int64_t)4 * RI * NVL * 64 - 1) / ((int64_t)4 * RI * NVL * 64));                         \
        if (silu_on && !beta) hipLaunchKernelGGL((rmsnorm_silu_flat_kernel<NVEC, NVL, RI, true, false>), dim3(nb), dim3(256), 0, st, xp, M, gamma, beta, yp); \
        else if (silu_on) hipLaunchKernelGGL((rmsnorm_silu_flat_kernel<NVEC, NVL, RI, true, true>), dim3(nb), dim3(256), 0, st, xp, M, gamma, beta, yp);      \
        else if (!beta) hipLaunchKernelGGL((rmsnorm_silu_flat_kernel<NVEC, NVL, RI, false, false>), dim3(nb), dim3(256), 0, st, xp, M, gamma, beta, yp);      \
        else hipLaunchKernelGGL((rmsnorm_silu_flat_kernel<NVEC, NVL, RI, false, true>), dim3(nb), dim3(256), 0, st, xp, M, gamma, beta, yp);                  \
    } while (0)
    const bool contiguous = ldx == C && ldy == C;
    if (grouped && flat && contiguous && nvec == 12) LAUNCH_RMS_F(12, 3, 2);
    else if (grouped && flat && contiguous && nvec == 24) LAUNCH_RMS_F(24, 3, 2);
    else if (grouped && flat && contiguous && nvec == 48) LAUNCH_RMS_F(48, 3, 2);
    else if (grouped && flat && contiguous && nvec == 20) LAUNCH_RMS_F(20, 5, 1);
    else if (grouped && flat && contiguous && nvec == 40) LAUNCH_RMS_F(40, 5, 1);
    else if (grouped && flat && contiguous && nvec == 80) LAUNCH_RMS_F(80, 5, 1);
    else if (grouped && nvl == 1) { LAUNCH_RMS_GN(1, 4); }
    else if (grouped && nvl == 3) { LAUNCH_RMS_GN(3, 2); }
    else if (grouped && nvl == 5) { LAUNCH_RMS_GN(5, 1); }
    else if (nvec <= 8) LAUNCH_RMS(8, 1);
    else if (nvec <= 16) LAUNCH_RMS(16, 1);
    else if (nvec <= 32) LAUNCH_RMS(32, 1);
    else if (nvec <= 64) LAUNCH_RMS(64, 1);
    else if (nvec <= 128) LAUNCH_RMS(64, 2);
    else if (nvec <= 256) LAUNCH_RMS(64, 4);
    else LAUNCH_RMS(64, 8);
#undef LAUNCH_RMS_F
#undef LAUNCH_RMS_GN
#undef LAUNCH_RMS_G
#undef LAUNCH_RMS
    YUME_CHECK_LAUNCH("vae_rmsnorm_silu");
    return YUME_OK;
}

extern "C" int yume_vae_dupup_add(const void* x, int64_t ldx, int64_t Tin, int64_t Hin, int64_t Win, int64_t Cin, void* y,
                                  int64_t ldy, int64_t To, int64_t Cout, int ft, int fs, int toff, void* stream) {
    YUME_REQUIRE(x && y, "vae_dupup_add: NULL pointer");
    YUME_REQUIRE(ft >= 1 && fs >= 1 && (Cout % 8) == 0 && (ldy % 8) == 0, "vae_dupup_add: bad shape");
    YUME_REQUIRE((Cout * ft * fs * fs) % Cin == 0, "vae_dupup_add: Cout*factor must be a multiple of Cin");
    YUME_REQUIRE(toff >= 0 && To + toff <= Tin * ft, "vae_dupup_add: frame range");
    const int64_t total = To * Hin * fs * Win * fs * (Cout / 8);
    {
        // the per-line form: repeats = Cout * F / Cin a power of two dividing F (Q = Cin / Cout in {1, 2, 4}); YUME_VAE_SHORTCUT_LINES=0: the form below
        const char* ev = getenv("YUME_VAE_SHORTCUT_LINES");
        const int64_t F = (int64_t)ft * fs * fs, repeats = Cout * F / Cin;
        const int64_t Ho = Hin * fs, Wo = Win * fs, cv = Cout / 8, lines = To * Ho;
        const bool pow2 = repeats > 0 && (repeats & (repeats - 1)) == 0;
        const int64_t Q = pow2 && (F % repeats) == 0 ? F / repeats : 0;
        if ((!ev || atoi(ev) != 0) && (Q == 1 || Q == 2 || Q == 4) && (ldx % 8) == 0 && Cin == Cout * Q && lines > 0 && lines < 65536 &&
            Wo * cv < (1ll << 30)) {
            int rep_shift = 0, cv_shift = -1;
            while ((1ll << rep_shift) < repeats) ++rep_shift;
            if ((cv & (cv - 1)) == 0) { cv_shift = 0; while ((1ll << cv_shift) < cv) ++cv_shift; }
            const unsigned gx = (unsigned)((Wo * cv + 1023) / 1024);      // four pixels-by-8-channels per thread
#define LAUNCH_DUP(QQ)                                                                                                                          \
    hipLaunchKernelGGL(dupup_add_lines_kernel<QQ>, dim3(gx, (unsigned)lines), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, ldx, \
                       (int)Hin, (int)Win, (unsigned short*)y, ldy, (int)Ho, (int)Wo, (int)cv, cv_shift, ft, fs, toff, rep_shift)
            if (Q == 1) LAUNCH_DUP(1); else if (Q == 2) LAUNCH_DUP(2); else LAUNCH_DUP(4);
#undef LAUNCH_DUP
            YUME_CHECK_LAUNCH("vae_dupup_add");
            return YUME_OK;
        }
    }
    hipLaunchKernelGGL(dupup_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                       ldx, (int)Tin, (int)Hin, (int)Win, (int)Cin, (unsigned short*)y, ldy, (int)To, (int)Cout, ft, fs, toff);
    YUME_CHECK_LAUNCH("vae_dupup_add");
    return YUME_OK;
}

extern "C" int yume_vae_avgdown_add(const void* x, int64_t ldx, int64_t Tin, int64_t Hin, int64_t Win, int64_t Cin, void* y,
                                    int64_t ldy, int64_t Cout, int ft, int fs, void* stream) {
    YUME_REQUIRE(x && y, "vae_avgdown_add: NULL pointer");
    YUME_REQUIRE(ft >= 1 && fs >= 1 && (Hin % fs) == 0 && (Win % fs) == 0, "vae_avgdown_add: H, W must be multiples of factor_s");
    YUME_REQUIRE((Cin * ft * fs * fs) % Cout == 0, "vae_avgdown_add: Cin*factor must be a multiple of Cout");
    const int64_t padt = (ft - Tin % ft) % ft;
    const int64_t total = ((Tin + padt) / ft) * (Hin / fs) * (Win / fs) * Cout;
    {
        // the per-line form: G = Cin * F / Cout dividing F (RR = F / G in {1, 2, 4, 8}), 8 output channels per thread
        const char* ev = getenv("YUME_VAE_SHORTCUT_LINES");
        const int64_t F = (int64_t)ft * fs * fs, G = Cin * F / Cout;
        const int64_t To = (Tin + padt) / ft, Ho = Hin / fs, Wo = Win / fs, cv = Cout / 8, lines = To * Ho;
        const int64_t RR = G > 0 && (F % G) == 0 ? F / G : 0;
        if ((!ev || atoi(ev) != 0) && (RR == 1 || RR == 2 || RR == 4 || RR == 8) && (Cout % 8) == 0 && (ldy % 8) == 0 && (ldx % 8) == 0 &&
            Cin * RR == Cout && lines > 0 && lines < 65536 && Wo * cv < (1ll << 30)) {
            int cv_shift = -1;
            if ((cv & (cv - 1)) == 0) { cv_shift = 0; while ((1ll << cv_shift) < cv) ++cv_shift; }
            const unsigned gx = (unsigned)((Wo * cv + 1023) / 1024);      // four pixels-by-8-channels per thread
#define LAUNCH_AVG(R_)                                                                                                                            \
    hipLaunchKernelGGL(avgdown_add_lines_kernel<R_>, dim3(gx, (unsigned)lines), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, ldx, \
                       (int)Hin, (int)Win, (unsigned short*)y, ldy, (int)Ho, (int)Wo, (int)cv, cv_shift, ft, fs, (int)G, (int)padt)
            if (RR == 1) LAUNCH_AVG(1); else if (RR == 2) LAUNCH_AVG(2); else if (RR == 4) LAUNCH_AVG(4); else LAUNCH_AVG(8);
#undef LAUNCH_AVG
            YUME_CHECK_LAUNCH("vae_avgdown_add");
            return YUME_OK;
        }
    }
    hipLaunchKernelGGL(avgdown_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                       ldx, (int)Tin, (int)Hin, (int)Win, (int)Cin, (unsigned short*)y, ldy, (int)Cout, ft, fs);
    YUME_CHECK_LAUNCH("vae_avgdown_add");
    return YUME_OK;
}

extern "C" int yume_softmax_rows(const float* S, int64_t lds_, int64_t R, int64_t n, float scale, void* P, int64_t ldp,
                                 void* stream) {
    YUME_REQUIRE(S && P, "softmax_rows: NULL pointer");
    YUME_REQUIRE(R > 0 && n > 0 && ldp >= n && lds_ >= n && n < (1ll << 30), "softmax_rows: bad shape");
    {
        const char* ev = getenv("YUME_VAE_SOFTMAX_REG");
        const bool reg_ok = (!ev || atoi(ev) != 0) && (lds_ % 4) == 0 && (ldp % 4) == 0 && ((uintptr_t)S % 16) == 0 && ((uintptr_t)P % 8) == 0 &&
                            lds_ >= (n + 3) / 4 * 4;
#define LAUNCH_SM(NV) hipLaunchKernelGGL(softmax_rows_reg_kernel<NV>, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, S, lds_, (int)n, scale, (unsigned short*)P, ldp)
        if (reg_ok && n <= 1024 * 2) LAUNCH_SM(2);
        else if (reg_ok && n <= 1024 * 4) LAUNCH_SM(4);
        else if (reg_ok && n <= 1024 * 8) LAUNCH_SM(8);
        else
            hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, S, lds_, (int)n, scale,
                               (unsigned short*)P, ldp);
#undef LAUNCH_SM
    }
    YUME_CHECK_LAUNCH("softmax_rows");
    return YUME_OK;
}

extern "C" int yume_vae_pack_input(const void* x, int in_bf16, int64_t C, int64_t T, int64_t H, int64_t W, int ps,
                                   const float* mul, const float* add, void* out, int64_t Cpad, void* stream) {
    YUME_REQUIRE(x && out, "vae_pack_input: NULL pointer");
    YUME_REQUIRE(ps >= 1 && (H % ps) == 0 && (W % ps) == 0 && Cpad >= C * ps * ps, "vae_pack_input: bad shape");
    YUME_REQUIRE((mul == nullptr) == (add == nullptr), "vae_pack_input: mul and add go together");
    const int64_t total = T * (H / ps) * (W / ps) * Cpad;
    hipStream_t st = (hipStream_t)stream;
    if (in_bf16)
        hipLaunchKernelGGL(pack_input_kernel<true>, dim3(grid_for(total)), dim3(256), 0, st, x, (int)C, (int)T, (int)H, (int)W, ps, mul, add, (unsigned short*)out, (int)Cpad);
    else
        hipLaunchKernelGGL(pack_input_kernel<false>, dim3(grid_for(total)), dim3(256), 0, st, x, (int)C, (int)T, (int)H, (int)W, ps, mul, add, (unsigned short*)out, (int)Cpad);
    YUME_CHECK_LAUNCH("vae_pack_input");
    return YUME_OK;
}

extern "C" int yume_vae_unpack_output(const void* x, int64_t ldx, int64_t T, int64_t H, int64_t W, int64_t Cv, int ps,
                                      const float* sub, const float* mul, float lo, float hi, float* out, void* stream) {
    YUME_REQUIRE(x && out, "vae_unpack_output: NULL pointer");
    YUME_REQUIRE(ps >= 1 && (Cv % (ps * ps)) == 0 && ldx >= Cv, "vae_unpack_output: bad shape");
    YUME_REQUIRE((mul == nullptr) == (sub == nullptr), "vae_unpack_output: sub and mul go together");
    const int64_t total = (Cv / (ps * ps)) * T * H * ps * W * ps;
    hipLaunchKernelGGL(unpack_output_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                       ldx, (int)T, (int)H, (int)W, (int)Cv, ps, sub, mul, lo, hi, out);
    YUME_CHECK_LAUNCH("vae_unpack_output");
    return YUME_OK;
}

template <bool TRUNC>
static int frames_u8_impl(const float* video, int64_t C, int64_t T, int64_t H, int64_t W, void* out, void* stream) {
    YUME_REQUIRE(video && out, "frames_u8: NULL pointer");
    YUME_REQUIRE(C >= 1 && C <= 4 && T > 0 && H > 0 && W > 0, "frames_u8: bad shape C=%lld T=%lld H=%lld W=%lld", (long long)C, (long long)T, (long long)H, (long long)W);
    const int64_t plane = T * H * W;
    YUME_REQUIRE((plane % 4) == 0, "frames_u8: T*H*W=%lld must be a multiple of 4", (long long)plane);
    YUME_REQUIRE(((uintptr_t)video % 16) == 0 && ((uintptr_t)out % 4) == 0, "frames_u8: pointer alignment");
    const int64_t nquad = plane / 4;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* o = (unsigned char*)out;
    switch ((int)C) {
        case 1: hipLaunchKernelGGL((frames_u8_kernel<1, TRUNC>), dim3(grid_for(nquad)), dim3(256), 0, st, video, plane, nquad, o); break;
        case 2: hipLaunchKernelGGL((frames_u8_kernel<2, TRUNC>), dim3(grid_for(nquad)), dim3(256), 0, st, video, plane, nquad, o); break;
        case 3: hipLaunchKernelGGL((frames_u8_kernel<3, TRUNC>), dim3(grid_for(nquad)), dim3(256), 0, st, video, plane, nquad, o); break;
        default: hipLaunchKernelGGL((frames_u8_kernel<4, TRUNC>), dim3(grid_for(nquad)), dim3(256), 0, st, video, plane, nquad, o); break;
    }
    YUME_CHECK_LAUNCH("frames_u8");
    return YUME_OK;
}

extern "C" int yume_frames_u8(const float* video, int64_t C, int64_t T, int64_t H, int64_t W, void* out, void* stream) {
    return frames_u8_impl<false>(video, C, T, H, W, out, stream);
}

extern "C" int yume_frames_u8_trunc(const float* video, int64_t C, int64_t T, int64_t H, int64_t W, void* out, void* stream) {
    return frames_u8_impl<true>(video, C, T, H, W, out, stream);
}

extern "C" int yume_softmax_bias_rows(const float* S, int64_t lds, int64_t strideS, int64_t H, int64_t n, const float* bias,
                                      int64_t ldb, void* P, int64_t ldp, int64_t strideP, void* stream) {
    YUME_REQUIRE(S && bias && P, "softmax_bias_rows: NULL pointer");
    YUME_REQUIRE(H > 0 && n > 0 && n <= 1024 && ldp >= n && ldp <= 1024 && lds >= n && ldb >= 2 * n - 1,
                 "softmax_bias_rows: bad shape H=%lld n=%lld lds=%lld ldp=%lld ldb=%lld", (long long)H, (long long)n, (long long)lds,
                 (long long)ldp, (long long)ldb);
    const int64_t rows = H * n;
    hipLaunchKernelGGL(softmax_bias_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, lds, strideS,
                       (int)n, bias, ldb, (unsigned short*)P, ldp, strideP, rows);
    YUME_CHECK_LAUNCH("softmax_bias_rows");
    return YUME_OK;
}
