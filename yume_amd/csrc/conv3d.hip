// conv3d.hip — implicit-GEMM convolutions of the causal 3D VAE on channels-last bf16 activations (gfx950).
//
//   out[(to,ho,wo), co] = bias[co] + sum_{dt,dh,dw,ci} in[(ti,hi,wi), ci] * W[co, ((dt*kh+dh)*kw+dw)*Cin + ci]
//   ti = to*st + dt - pt   (ti < 0 reads the 2-frame causal cache: frame 2+ti, or zeros without a cache)
//   hi = ho*sh + dh - ph , wi = wo*sw + dw - pw   (outside the frame -> 0); with `ups` the spatial gather
//   addresses a nearest-2x-upsampled view of the input (hi>>1, wi>>1), folding nn.Upsample into the conv.
//
// This is gemm_core's pipeline with M = To*Ho*Wo, N = Cout, K = kt*kh*kw*Cin (padded to 64) and an A loader that
// computes, per 16-byte chunk (8 channels), the address of the tap it belongs to; out-of-image / padded taps
// are fetched from a caller-provided zero page so the LDS-DMA stays unconditional.
// One kernel covers every conv in the VAE: CausalConv3d 3x3x3 and (3,1,1) (with/without temporal stride 2),
// Conv2d 3x3 after nearest-exact x2, Conv2d 3x3 stride 2 after ZeroPad2d((0,1,0,1)), and 1x1x1.
// Roofline: MFMA (bf16 dense); algorithmic work 2*M*Cout*Cin*kt*kh*kw flop per launch.
#include "gemm_core.hpp"
#include "conv_w4.hpp"
#include "conv_halo.hpp"
#include "conv_halo_n.hpp"
#include "conv_in.hpp"

using namespace gemm_core;

namespace {

struct ConvA {
    __device__ __forceinline__ void batch_offset(int) {}
    const unsigned short* x;       // [Tin, Hin, Win, ldc]
    const unsigned short* cache;   // [2, Hin, Win, ldc] or nullptr
    const unsigned short* zero;    // >= 16 bytes of zeros
    int64_t ldc;
    int Tin, Hin, Win, Cin;
    int To, Ho, Wo, M;
    int kt, kh, kw, st, sh, sw, pt, ph, pw, ups;
    // per-thread state
    int t0[4], h0[4], w0[4];
    int cin, dt, dh, dw;

    __device__ __forceinline__ void init(int m0, int tid, int rpr = 32, int kshift = 0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            int m = m0 + rr * rpr + (tid >> 3);
            m = m < M ? m : M - 1;
            const int wo = m % Wo;
            const int ho = (m / Wo) % Ho;
            const int to = m / (Wo * Ho);
            t0[rr] = to * st - pt;
            h0[rr] = ho * sh - ph;
            w0[rr] = wo * sw - pw;
        }
        const int lc = (tid & 7) ^ (((tid >> 3) >> kshift) & 7);   // logical 16-byte chunk of this thread (same for its 4 rows)
        const int k = lc * 8;
        int tap = k / Cin;
        cin = k - tap * Cin;
        dw = tap % kw;
        dh = (tap / kw) % kh;
        dt = tap / (kw * kh);
    }
    __device__ __forceinline__ const unsigned short* src(int rr, int /*kt_unused*/) const {
        const int ti = t0[rr] + dt;
        int hi = h0[rr] + dh, wi = w0[rr] + dw;
        bool ok = dt < kt;
        if (ups) {
            ok = ok && hi >= 0 && hi < 2 * Hin && wi >= 0 && wi < 2 * Win;
            hi >>= 1;
            wi >>= 1;
        } else {
            ok = ok && hi >= 0 && hi < Hin && wi >= 0 && wi < Win;
        }
        const unsigned short* base = x;
        int tf = ti;
        if (ti < 0) {
            base = cache;
            tf = ti + 2;
            ok = ok && cache != nullptr && tf >= 0;
        }
        if (!ok) return zero;
        const int64_t pos = ((int64_t)tf * Hin + hi) * Win + wi;
        return base + pos * ldc + cin;
    }
    __device__ __forceinline__ void advance() {
        cin += BK;
        while (cin >= Cin) {
            cin -= Cin;
            if (++dw == kw) {
                dw = 0;
                if (++dh == kh) { dh = 0; ++dt; }
            }
        }
    }
    __device__ __forceinline__ int wk(int kt, bool /*behind*/) const { return kt * BK; }      // K walked in the weight's own order
};


// Fast path for Cin % 64 == 0 without the folded upsample: a whole 64-wide K tile then lies inside ONE tap, so the tap
// state (dt, dh, dw, channel offset) is wave-uniform (scalar registers) and the per-row work per K tile shrinks to a
// validity-bit test, one compare and a 64-bit add: every row keeps the address of its tap-(0,0,0) source element and a
// 27-bit mask of the spatially valid taps; frames before the chunk (ti < 0) are the same offset from the cache base.
struct ConvAFast {
    __device__ __forceinline__ void batch_offset(int) {}
    const unsigned short* x;
    const unsigned short* cache;
    const unsigned short* zero;
    int64_t ldc;
    int Tin, Hin, Win, Cin;
    int To, Ho, Wo, M;
    int kt, kh, kw, st, sh, sw, pt, ph, pw, ups;
    // per-thread state
    const unsigned short* xbase[4];
    unsigned mask[4];
    int t0[4];
    // uniform state: position (dt, dh, cin, dw) of the K walk and the element offset of that K tile inside a weight row (and of the tile before)
    int cin, dt, dh, dw, tap;
    int wcur, wprev;
    int k_order;      // 2 (default): K walked (dt, channel tile, dh, dw); 1: (dt, dh, channel tile, dw); 0: the weight's own order (env YUME_CONV_KORDER, A/B)

    __device__ __forceinline__ void init(int m0, int tid, int rpr = 32, int kshift = 0) {
        const int lc = (tid & 7) ^ (((tid >> 3) >> kshift) & 7);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            int m = m0 + rr * rpr + (tid >> 3);
            m = m < M ? m : M - 1;
            const int wo = m % Wo;
            const int ho = (m / Wo) % Ho;
            const int to = m / (Wo * Ho);
            const int tt = to * st - pt, hh = ho * sh - ph, ww = wo * sw - pw;
            t0[rr] = tt;
            xbase[rr] = x + (((int64_t)tt * Hin + hh) * Win + ww) * ldc + lc * 8;   // may point outside; only used when valid
            unsigned mk = 0;
            for (int a = 0; a < kh; ++a)
                for (int b = 0; b < kw; ++b)
                    if (hh + a >= 0 && hh + a < Hin && ww + b >= 0 && ww + b < Win) mk |= 1u << (a * kw + b);
            mask[rr] = mk;
        }
        cin = 0; dt = 0; dh = 0; dw = 0; tap = 0;
        wcur = 0; wprev = 0;
    }
    __device__ __forceinline__ const unsigned short* src(int rr, int /*kt_unused*/) const {
        // uniform: offset of tap (dt,dh,dw) and of the cache frames relative to x
        const int64_t soff = (((int64_t)dt * Hin + dh) * Win + dw) * ldc + cin;
        const int ti = t0[rr] + dt;
        const bool neg = ti < 0;
        bool ok = dt < kt && ((mask[rr] >> (dh * kw + dw)) & 1u);
        const unsigned short* ptr = xbase[rr] + soff;
        if (neg) {
            ok = ok && cache != nullptr;
            ptr += (cache - x) + (int64_t)2 * Hin * Win * ldc;
        }
        return ok ? ptr : zero;
    }
    // K is NOT walked in the weight's own (dt, dh, dw, channel) order. order 2 (default): (dt, channel tile, dh, dw) — the kh*kw taps of
    // one frame and one 64-channel slice read the same few input rows shifted by a position / a row, so the 32 workgroups sharing an L2
    // re-read for 9 consecutive K tiles what the first of them fetched (working set: their ~15 rows x 128 B per position ~ 1.2 MB of the
    // 4 MiB L2) instead of coming back to it Cin/64 tiles (4+ MB of other traffic) later. order 1: (dt, dh, channel tile, dw).
    // The weight tile that belongs to the position is addressed through wk() (same products, another summation order).
    __device__ __forceinline__ void advance() {
        wprev = wcur;
        if (k_order == 0) {
            wcur += BK;
            cin += BK;
            if (cin >= Cin) {
                cin = 0;
                if (++dw == kw) {
                    dw = 0;
                    if (++dh == kh) { dh = 0; ++dt; }
                }
            }
            return;
        }
        if (++dw < kw) {
            wcur += Cin;
            return;
        }
        dw = 0;
        wcur -= (kw - 1) * Cin;
        if (k_order == 1) {
            cin += BK;
            wcur += BK;
            if (cin >= Cin) {
                cin = 0;
                wcur += (kw - 1) * Cin;           // = start of the next row of taps: ((dt*kh + dh + 1) * kw) * Cin
                if (++dh == kh) { dh = 0; ++dt; }
            }
            return;
        }
        if (++dh < kh) {
            wcur += kw * Cin;
            return;
        }
        dh = 0;
        wcur -= (kh - 1) * kw * Cin;
        cin += BK;
        wcur += BK;
        if (cin >= Cin) {
            cin = 0;
            ++dt;
            wcur += (kh * kw - 1) * Cin;          // = ((dt + 1) * kh * kw) * Cin
        }
    }
    __device__ __forceinline__ int wk(int /*kt*/, bool behind) const { return behind ? wprev : wcur; }
};

}  // namespace

extern "C" int yume_conv3d_cl(const void* x, const void* cache, int64_t ldc, int64_t Tin, int64_t Hin, int64_t Win,
                              int64_t Cin, const void* W, int64_t ldw, const float* bias, int64_t Cout, int kt, int kh,
                              int kw, int st, int sh, int sw, int pt, int ph, int pw, int ups, int64_t To, int64_t Ho,
                              int64_t Wo, int epi, void* out, int64_t ldo, const void* add, int64_t ldadd,
                              const void* zero_page, void* stream) {
    const int variant = 0;
    YUME_REQUIRE(x && W && out && zero_page, "conv3d_cl: NULL pointer");
    if (epi == YUME_CONV_EPI_RMS_SILU) {
        // out = SiLU(RMS_norm(conv + bias) * gamma), `add` = the fp32 gamma[Cout]: fused into the epilogue where one workgroup holds a position's
        // whole channel row (conv_halo_n.hpp, 96 / 160 output channels); elsewhere the plain convolution and the norm kernel in place behind it
        YUME_REQUIRE(add != nullptr, "conv3d_cl: the RMS_SILU epilogue needs gamma in `add`");
        static const bool fuse = [] { const char* v = getenv("YUME_CONV_FUSE_NORM"); return !v || atoi(v) != 0; }();
        if (!fuse || !conv_halo_n::applies(Cin, Cout, kt, kh, kw, st, sh, sw, pt, ph, pw, ups, Hin, Win, Ho, Wo, ldc, ldo, ldw, epi, 0)) {
            int rc = yume_conv3d_cl(x, cache, ldc, Tin, Hin, Win, Cin, W, ldw, bias, Cout, kt, kh, kw, st, sh, sw, pt, ph, pw, ups, To, Ho, Wo,
                                    YUME_EPI_BF16, out, ldo, nullptr, 0, zero_page, stream);
            if (rc != YUME_OK) return rc;
            return yume_vae_rmsnorm_silu(out, ldo, To * Ho * Wo, Cout, (const float*)add, nullptr, 1, out, ldo, stream);
        }
    }
    YUME_REQUIRE(Tin > 0 && Hin > 0 && Win > 0 && Cin > 0 && Cout > 0 && To > 0 && Ho > 0 && Wo > 0, "conv3d_cl: empty shape");
    YUME_REQUIRE((Cin % 8) == 0 && (ldc % 8) == 0 && ldc >= Cin, "conv3d_cl: Cin=%lld and ldc=%lld must be multiples of 8", (long long)Cin, (long long)ldc);
    YUME_REQUIRE((Cout % 4) == 0 && (ldo % 4) == 0, "conv3d_cl: Cout and ldo must be multiples of 4");
    YUME_REQUIRE(ldo >= (epi == 17 ? Cout / 2 : Cout), "conv3d_cl: ldo=%lld is smaller than the %lld output channels of a row", (long long)ldo,
                 (long long)(epi == 17 ? Cout / 2 : Cout));
    YUME_REQUIRE(kt >= 1 && kh >= 1 && kw >= 1 && st >= 1 && sh >= 1 && sw >= 1 && pt >= 0 && pt <= 2, "conv3d_cl: bad kernel geometry");
    const int64_t Ktrue = (int64_t)kt * kh * kw * Cin;
    const int64_t Kp = (Ktrue + BK - 1) / BK * BK;
    YUME_REQUIRE(ldw >= Kp && (ldw % 8) == 0, "conv3d_cl: weight rows must hold K padded to 64 (%lld), ldw=%lld", (long long)Kp, (long long)ldw);
    // the last input frame/row/col a valid tap may touch must exist
    YUME_REQUIRE((To - 1) * st + (kt - 1) - pt < Tin, "conv3d_cl: temporal extent exceeds the input (To=%lld Tin=%lld)", (long long)To, (long long)Tin);
    const int64_t M = To * Ho * Wo;
    YUME_REQUIRE(M < (1ll << 31), "conv3d_cl: too many output positions");
    YUME_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)zero_page % 16) == 0 &&
                 (cache == nullptr || ((uintptr_t)cache % 16) == 0), "conv3d_cl: pointer alignment");
    Problem p;
    p.W = (const unsigned short*)W; p.ldw = ldw;
    p.M = (int)M; p.N = (int)Cout; p.K = (int)Kp;
    p.tiles_m = (int)((M + BM - 1) / BM);
    p.tiles_n = (int)((Cout + BN - 1) / BN);
    p.group_m = 8;
    p.bsW = 0; p.bsO = 0;
    ConvA al = {};
    al.x = (const unsigned short*)x; al.cache = (const unsigned short*)cache; al.zero = (const unsigned short*)zero_page;
    al.ldc = ldc;
    al.Tin = (int)Tin; al.Hin = (int)Hin; al.Win = (int)Win; al.Cin = (int)Cin;
    al.To = (int)To; al.Ho = (int)Ho; al.Wo = (int)Wo; al.M = (int)M;
    al.kt = kt; al.kh = kh; al.kw = kw; al.st = st; al.sh = sh; al.sw = sw; al.pt = pt; al.ph = ph; al.pw = pw; al.ups = ups;
    Epilogue e = {};
    e.bias = bias;
    e.out = out; e.ldo = ldo;
    e.add = (const unsigned short*)add; e.ldadd = ldadd;
    e.hw = (int)(Ho * Wo);
    hipStream_t s = (hipStream_t)stream;
    const bool big = use_256(p, variant, true);
    if (conv_in::applies(Cin, Cout, kt, kh, kw, st, sh, sw, pt, ph, pw, ups, Hin, Win, Ho, Wo, ldc, ldo, ldw, epi)) {
        // the encoders' first convolution (8 -> 96, 16 -> 160 channels): weights resident in registers, halo tile in LDS (conv_in.hpp)
        conv_in::Params ip;
        ip.x = al.x; ip.cache = al.cache; ip.w = (const unsigned short*)W; ip.bias = bias; ip.out = (unsigned short*)out;
        ip.ldw = ldw; ip.ldo = ldo;
        ip.Tin = (int)Tin; ip.H = (int)Hin; ip.W = (int)Win; ip.To = (int)To; ip.cout = (int)Cout;
        static const bool log_on = [] { const char* v = getenv("YUME_CONV_LOG"); return v && atoi(v) != 0; }();
        if (log_on) fprintf(stderr, "[conv3d_cl] conv_in M=%lld Cin=%lld Cout=%lld k=%dx%dx%d\n", (long long)M, (long long)Cin, (long long)Cout, kt, kh, kw);
        int rc = 0;
        if (Cin == 8) {
            rc = conv_in::launch<8, 6>(ip, To, Ho, Wo, s);
        } else {
            for (int ch = 0; ch < 2 && rc == 0; ++ch) {          // 160 output channels = two launches of 80 (280 weight registers per lane each)
                conv_in::Params q = ip;
                q.w = ip.w + (int64_t)ch * 80 * ldw;
                q.bias = bias ? bias + ch * 80 : nullptr;
                q.out = ip.out + ch * 80;
                q.cout = 80;
                rc = conv_in::launch<16, 5>(q, To, Ho, Wo, s);
            }
        }
        YUME_REQUIRE(rc == 0, "conv3d_cl: too many tiles");
        YUME_CHECK_LAUNCH("conv3d_cl");
        return YUME_OK;
    }
    if (conv_halo::applies(Cin, Cout, kt, kh, kw, st, sh, sw, pt, ph, pw, ups, Hin, Win, Ho, Wo, ldc, ldo, ldw, epi)) {
        // few output channels at full resolution (the decoder's head): halo tile in LDS, the 9 in-plane taps out of it (conv_halo.hpp)
        conv_halo::Params hp;
        hp.x = al.x; hp.cache = al.cache; hp.w = (const unsigned short*)W; hp.bias = bias; hp.out = (unsigned short*)out;
        hp.ldc = ldc; hp.ldw = ldw; hp.ldo = ldo;
        hp.Tin = (int)Tin; hp.H = (int)Hin; hp.W = (int)Win; hp.C = (int)Cin; hp.To = (int)To; hp.cout = (int)Cout;
        hp.tiles_w = (int)((Wo + conv_halo::TW - 1) / conv_halo::TW);
        hp.tiles_h = (int)((Ho + conv_halo::TH - 1) / conv_halo::TH);
        const int64_t nt = To * hp.tiles_h * hp.tiles_w;
        YUME_REQUIRE(nt < (1ll << 31), "conv3d_cl: too many tiles");
        hipLaunchKernelGGL(conv_halo::conv_halo16_kernel<0>, dim3((unsigned)nt), dim3(256), 0, s, hp);
        YUME_CHECK_LAUNCH("conv3d_cl");
        return YUME_OK;
    }
    if (conv_halo_n::applies(Cin, Cout, kt, kh, kw, st, sh, sw, pt, ph, pw, ups, Hin, Win, Ho, Wo, ldc, ldo, ldw, epi, ldadd)) {
        // 96 / 160-channel levels (Cin in whole 32-channel slices, one N tile of Cout): halo tile + weight ring in LDS (conv_halo_n.hpp)
        YUME_REQUIRE(epi != YUME_CONV_EPI_ADD || add != nullptr, "conv3d_cl: ADD epilogue needs an addend");
        conv_halo_n::Params hp;
        hp.x = al.x; hp.cache = al.cache; hp.w = (const unsigned short*)W; hp.bias = bias; hp.out = (unsigned short*)out;
        hp.add = epi == YUME_CONV_EPI_RMS_SILU ? nullptr : (const unsigned short*)add;
        hp.gamma = epi == YUME_CONV_EPI_RMS_SILU ? (const float*)add : nullptr;
        hp.ldc = ldc; hp.ldw = ldw; hp.ldo = ldo; hp.ldadd = ldadd;
        hp.Tin = (int)Tin; hp.H = (int)Hin; hp.W = (int)Win; hp.C = (int)Cin; hp.To = (int)To; hp.cout = (int)Cout; hp.kt = kt; hp.pt = pt;
        static const bool log_on = [] { const char* v = getenv("YUME_CONV_LOG"); return v && atoi(v) != 0; }();
        if (log_on) fprintf(stderr, "[conv3d_cl] halo_n M=%lld Cin=%lld Cout=%lld k=%dx%dx%d\n", (long long)M, (long long)Cin, (long long)Cout, kt, kh, kw);
        const int addep = epi == YUME_CONV_EPI_ADD ? 1 : epi == YUME_CONV_EPI_RMS_SILU ? 2 : 0;
        // 96 channels: two waves per SIMD (8 waves x 4 m-tiles of one workgroup; measured 3-5 % ahead of 4 waves x 8 m-tiles in the first build,
        // whose unrolled tap loop no longer fits the register file without scratch traffic: not instantiated); 160 channels / the 16-channel head:
        // one wave per SIMD (profiles/r6_conv_halo_n_ablations.log)
        int rc = 0;
        const int inst = conv_halo_n::instance(Cin, Cout, ups);
        // an N tile is the instance's 96 / 160 channels: a convolution that widens the level (96 -> 192, 160 -> 320) is one launch per tile of
        // output channels (the input is read once per launch: two launches at 1.1 PF beat one on the generic-loader kernel at 0.44)
        const int nchunk = (inst == 96 || inst == 160) ? (int)(Cout / inst) : 1;
        for (int ch = 0; ch < nchunk && rc == 0; ++ch) {
            conv_halo_n::Params q = hp;
            if (nchunk > 1) {
                q.w = hp.w + (int64_t)ch * inst * ldw;
                q.bias = bias ? bias + ch * inst : nullptr;
                q.out = hp.out + ch * inst;
                q.add = hp.add ? hp.add + ch * inst : nullptr;
                q.cout = inst;
            }
            if (ups)
                rc = addep ? -2 : conv_halo_n::launch_inst<6, 4, 8, 64, 8, true>(q, To, Ho, Wo, 0, s);
            else if (inst == 16)
                rc = conv_halo_n::launch_inst<1, 8, 8, 64, 4>(q, To, Ho, Wo, addep, s);
            else if (inst == 96)
                rc = conv_halo_n::launch_inst<6, 4, 8, 64, 8>(q, To, Ho, Wo, addep, s);
            else
                rc = conv_halo_n::launch_inst<10, 4, 4, 64, 4>(q, To, Ho, Wo, addep, s);
        }
        YUME_REQUIRE(rc == 0, "conv3d_cl: too many tiles (or an upsample convolution with a shortcut)");
        YUME_CHECK_LAUNCH("conv3d_cl");
        return YUME_OK;
    }
    {
        // stride-1 convolutions over whole-tile frames: the one-wave-per-SIMD pipeline (conv_w4.hpp); YUME_CONV_W4=0 keeps the 8-wave kernel
        gemm_w4::ConvW4 cv;
        cv.x = al.x; cv.cache = al.cache; cv.ldc = ldc;
        cv.Tin = (int)Tin; cv.Hin = (int)Hin; cv.Win = (int)Win; cv.Cin = (int)Cin; cv.To = (int)To; cv.Ho = (int)Ho; cv.Wo = (int)Wo;
        cv.kt = kt; cv.kh = kh; cv.kw = kw; cv.pt = pt; cv.ph = ph; cv.pw = pw; cv.ups = ups ? 1 : 0;
        const int e2 = epi == YUME_CONV_EPI_ADD ? EPI_BF16_ADD : epi == YUME_CONV_EPI_TSPLIT ? EPI_BF16_TSPLIT : epi;
        Problem p2 = p;
        p2.tiles_m = (int)((M + 255) / 256);
        p2.tiles_n = (int)((Cout + 255) / 256);
        // (r5, measured and not kept: launches too small to fill the chip — the first-chunk passes, 56 tiles with K = 27 x 1024 — on this
        // pipeline instead of the 128x128 kernel: decode 386.4-387.9 vs 387.3-388.2 ms, both K loops run at memory latency there;
        // profiles/r5_vae_small_launch_ab.log. YUME_CONV_LOG=1 prints the kernel every call takes.)
        // (N = 384 / 320 / 640 channels: 75 / 62 / 83 % of their 256-wide N tiles is real work, and the pipeline is worth 2x the gathering
        // 128^2 kernel per padded flop — YUME_CONV_W4_WORTH overrides the 2.0 for A/B runs, 1.25 = the r4 rule)
        static const double w4_worth = [] { const char* v = getenv("YUME_CONV_W4_WORTH"); const double d = v ? atof(v) : 2.0; return d > 0.5 ? d : 2.0; }();
        // (at Cout <= 128 the worth-2.0 query is an exact tie whose outcome follows the parity of ceil(M / 128) — a layer could
        // flip kernels between a grouped and a one-latent pass; those widths now have their own kernel (conv_halo_n.hpp) or stay on the
        // gathering kernel: the relaxed query needs at least 160 real channels in its 256-wide tile)
        const bool big_w4 = big || (Cout >= 160 && use_256(p, variant, true, w4_worth));
        if (big_w4 && gemm_w4::conv_w4_applies(p2, cv, st, sh, sw, ups, e2) && (epi != YUME_CONV_EPI_ADD || (add != nullptr && (ldadd % 4) == 0))) {
            static const bool log_on = [] { const char* v = getenv("YUME_CONV_LOG"); return v && atoi(v) != 0; }();
            if (log_on) fprintf(stderr, "[conv3d_cl] w4   M=%lld Cin=%lld Cout=%lld k=%dx%dx%d ups=%d tiles=%d\n", (long long)M, (long long)Cin, (long long)Cout, kt, kh, kw, (int)ups, p2.tiles_m * p2.tiles_n);
            return gemm_w4::launch_conv_w4(e2, p, cv, e, s, "conv3d_cl");
        }
        {
            static const bool log_on = [] { const char* v = getenv("YUME_CONV_LOG"); return v && atoi(v) != 0; }();
            if (log_on) fprintf(stderr, "[conv3d_cl] %s M=%lld Cin=%lld Cout=%lld k=%dx%dx%d stride=%d,%d,%d ups=%d\n", big ? "g256" : "g128", (long long)M, (long long)Cin, (long long)Cout, kt, kh, kw, st, sh, sw, (int)ups);
        }
    }
    if ((Cin % BK) == 0 && !ups && kh * kw <= 32) {
        ConvAFast af = {};
        af.x = al.x; af.cache = al.cache; af.zero = al.zero; af.ldc = al.ldc;
        af.Tin = al.Tin; af.Hin = al.Hin; af.Win = al.Win; af.Cin = al.Cin;
        af.To = al.To; af.Ho = al.Ho; af.Wo = al.Wo; af.M = al.M;
        af.kt = kt; af.kh = kh; af.kw = kw; af.st = st; af.sh = sh; af.sw = sw; af.pt = pt; af.ph = ph; af.pw = pw; af.ups = 0;
        { static const int ko = [] { const char* v = getenv("YUME_CONV_KORDER"); return v ? atoi(v) : 2; }(); af.k_order = (ko == 0 || ko == 1) ? ko : 2; }
        switch (epi) {
            case YUME_EPI_BF16: return big ? launch256<YUME_EPI_BF16>(p, af, e, s, "conv3d_cl", 0) : launch<YUME_EPI_BF16>(p, af, e, s, "conv3d_cl");
            case YUME_EPI_F32: return big ? launch256<YUME_EPI_F32>(p, af, e, s, "conv3d_cl", 0) : launch<YUME_EPI_F32>(p, af, e, s, "conv3d_cl");
            case YUME_CONV_EPI_ADD:
                YUME_REQUIRE(add != nullptr && (ldadd % 4) == 0, "conv3d_cl: ADD epilogue needs an addend with ldadd %% 4 == 0");
                return big ? launch256<EPI_BF16_ADD>(p, af, e, s, "conv3d_cl", 0) : launch<EPI_BF16_ADD>(p, af, e, s, "conv3d_cl");
            case YUME_CONV_EPI_TSPLIT:
                YUME_REQUIRE((Cout % 8) == 0, "conv3d_cl: TSPLIT needs an even channel split");
                return big ? launch256<EPI_BF16_TSPLIT>(p, af, e, s, "conv3d_cl", 0) : launch<EPI_BF16_TSPLIT>(p, af, e, s, "conv3d_cl");
            default:
                yume_set_error("conv3d_cl: unknown epilogue %d", epi);
                return YUME_EINVAL;
        }
    }
    switch (epi) {
        case YUME_EPI_BF16: return big ? launch256<YUME_EPI_BF16>(p, al, e, s, "conv3d_cl", 0) : launch<YUME_EPI_BF16>(p, al, e, s, "conv3d_cl");
        case YUME_EPI_F32: return big ? launch256<YUME_EPI_F32>(p, al, e, s, "conv3d_cl", 0) : launch<YUME_EPI_F32>(p, al, e, s, "conv3d_cl");
        case YUME_CONV_EPI_ADD:
            YUME_REQUIRE(add != nullptr && (ldadd % 4) == 0, "conv3d_cl: ADD epilogue needs an addend with ldadd %% 4 == 0");
            return big ? launch256<EPI_BF16_ADD>(p, al, e, s, "conv3d_cl", 0) : launch<EPI_BF16_ADD>(p, al, e, s, "conv3d_cl");
        case YUME_CONV_EPI_TSPLIT:
            YUME_REQUIRE((Cout % 8) == 0, "conv3d_cl: TSPLIT needs an even channel split");
            return big ? launch256<EPI_BF16_TSPLIT>(p, al, e, s, "conv3d_cl", 0) : launch<EPI_BF16_TSPLIT>(p, al, e, s, "conv3d_cl");
        default:
            yume_set_error("conv3d_cl: unknown epilogue %d", epi);
            return YUME_EINVAL;
    }
}
