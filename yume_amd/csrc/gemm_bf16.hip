// gemm_bf16.hip — C-ABI of the dense bf16 MFMA GEMM (pipeline and epilogues: gemm_core.hpp).
// Roofline: MFMA (bf16 dense 2.5 PFLOP/s); algorithmic work 2*M*N*K flop per launch.
#include "gemm_core.hpp"
#include "gemm_w4.hpp"

using namespace gemm_core;

namespace {
// variant 0 (automatic) takes the one-wave-per-SIMD kernel (gemm_w4.hpp) wherever it took the 8-wave 256x256 kernel; YUME_GEMM_W4=0 keeps the latter
bool w4_auto() {
    static const bool on = [] { const char* v = getenv("YUME_GEMM_W4"); return !v || atoi(v) != 0; }();
    return on;
}
}  // namespace

extern "C" int64_t yume_gemm_workspace_bytes(void) { return gemm_w4::sk_workspace_bytes(gemm_w4::SK_MAX_SLOTS); }

extern "C" int yume_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int64_t M,
                              int64_t N, int64_t K, int epi, void* out, int64_t ldo, const float* gate,
                              int64_t gate_stride, const int32_t* row_idx, void* outT, int64_t ldt, int64_t n_split,
                              int variant, void* stream) {
    return yume_gemm_bf16_ws(A, lda, W, ldw, bias, M, N, K, epi, out, ldo, gate, gate_stride, row_idx, outT, ldt, n_split, variant, nullptr, 0, stream);
}

// ... with caller-owned scratch for the stream-K tail of the one-wave-per-SIMD kernel (yume_gemm_workspace_bytes() bytes, 16-byte aligned,
// ZERO when first handed over; a launch leaves it zero again; launches sharing one scratch must be ordered on one stream). NULL / too small:
// the schedules without it (whole tiles, row split).
extern "C" int yume_gemm_bf16_ws(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int64_t M,
                                 int64_t N, int64_t K, int epi, void* out, int64_t ldo, const float* gate,
                                 int64_t gate_stride, const int32_t* row_idx, void* outT, int64_t ldt, int64_t n_split,
                                 int variant, void* workspace, int64_t workspace_bytes, void* stream) {
    YUME_REQUIRE(A && W && out, "gemm_bf16: NULL pointer");
    YUME_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_bf16: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    YUME_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "gemm_bf16: dimension too large");
    YUME_REQUIRE((K % BK) == 0, "gemm_bf16: K=%lld must be a multiple of %d", (long long)K, BK);
    YUME_REQUIRE((N % 4) == 0, "gemm_bf16: N=%lld must be a multiple of 4", (long long)N);
    YUME_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0, "gemm_bf16: lda/ldw must be multiples of 8 (16-byte rows)");
    YUME_REQUIRE((ldo % 4) == 0, "gemm_bf16: ldo must be a multiple of 4");
    YUME_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_bf16: pointers must be 16-byte aligned");
    Problem p;
    p.W = (const unsigned short*)W; p.ldw = ldw;
    p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.tiles_m = (int)((M + BM - 1) / BM);
    p.tiles_n = (int)((N + BN - 1) / BN);
    p.group_m = 8;
    p.bsW = 0; p.bsO = 0;
    // the stream-K tail (gemm_w4.hpp) replaces the row split where its scratch is there and the plan takes the shape
    bool sk = false;
    if (variant == 0 && workspace != nullptr && w4_auto() && epi != YUME_EPI_BF16_GEGLU &&
        use_256(p, 0, epi != YUME_EPI_BF16_SPLITT || (n_split % 256) == 0) && gemm_w4::w4_applies(p, lda, epi) &&
        (epi != YUME_EPI_BF16_SPLITT || (ldt % 8) == 0)) {
        Problem q = p;
        q.tiles_m = (int)((M + 255) / 256);
        q.tiles_n = (int)((N + 255) / 256);
        gemm_w4::w4_sk_plan(q, workspace, workspace_bytes);
        sk = q.sk_wgs != 0;
    }
    if (!sk && variant == 0 && epi != YUME_EPI_BF16_GEGLU && use_256(p, 0, epi != YUME_EPI_BF16_SPLITT || (n_split % 256) == 0)) {
        // Row split (same idea as the attention's query split): when the 256x256 tiling leaves a last round that is at
        // most ~1/3 full (ffn.0 at L = 9460: 2072 tiles = 8 rounds + 24; QKV: 5 rounds + 52), run whole rounds of
        // M-tiles on the 256x256 kernel and the remaining rows (a few hundred) on the 128x128 kernel.
        const int64_t tn = (N + 255) / 256, tm = (M + 255) / 256, T = tm * tn, R = T % 256, full = T - R;
        const int64_t m_main = full / tn, M_main = m_main * 256, rem = M - M_main;
        if (R > 0 && R <= 96 && full >= 256 && m_main >= 1 && rem > 0 && rem <= 1024) {
            const int64_t osz = (epi == YUME_EPI_F32 || epi == YUME_EPI_RESID) ? 4 : 2;
            auto remainder = [&](void* s2) {
                return yume_gemm_bf16(reinterpret_cast<const unsigned short*>(A) + M_main * lda, lda, W, ldw, bias, rem, N, K, epi,
                                      reinterpret_cast<char*>(out) + M_main * ldo * osz, ldo, gate, gate_stride,
                                      row_idx ? row_idx + M_main : nullptr,
                                      outT ? reinterpret_cast<unsigned short*>(outT) + M_main : nullptr, ldt, n_split, 1, s2);
            };
            // (r2/r3 experiment, removed in r4: the remainder on a library-owned side stream, fork / join by events, bought 30 us of overlap
            // and paid most of it back in the two event hops — and made the library own a stream. Both launches go to the caller's stream.)
            int rc = yume_gemm_bf16(A, lda, W, ldw, bias, M_main, N, K, epi, out, ldo, gate, gate_stride, row_idx, outT, ldt, n_split,
                                    w4_auto() ? 3 : 2, stream);
            if (rc != YUME_OK) return rc;
            return remainder(stream);
        }
    }
    PlainA al;
    al.A = (const unsigned short*)A; al.lda = lda; al.M = (int)M; al.bsA = 0;
    Epilogue e = {};
    e.bias = bias;
    e.out = out; e.ldo = ldo;
    e.gate = gate; e.gate_stride = gate_stride; e.row_idx = row_idx;
    e.outT = (unsigned short*)outT; e.ldt = ldt; e.n_split = (int)n_split;
    hipStream_t st = (hipStream_t)stream;
    const bool split_ok = epi != YUME_EPI_BF16_SPLITT || (n_split % 256) == 0;
    const bool big = use_256(p, variant == 3 ? 2 : variant, split_ok);
    // variant 3: the one-wave-per-SIMD 256x256 kernel where it applies (else as variant 2); variant 0 takes it wherever it took the 8-wave kernel
    const bool w4 = (variant == 3 || (variant == 0 && big && w4_auto())) && split_ok && gemm_w4::w4_applies(p, lda, epi) &&
                    (epi != YUME_EPI_BF16_SPLITT || (ldt % 8) == 0);
#define YUME_GO(E) (w4 ? gemm_w4::launch_w4(E, p, al, e, st, "gemm_bf16", variant == 0 ? workspace : nullptr, workspace_bytes) : big ? launch256<E>(p, al, e, st, "gemm_bf16") : launch<E>(p, al, e, st, "gemm_bf16"))
    switch (epi) {
        case YUME_EPI_BF16: return YUME_GO(YUME_EPI_BF16);
        case YUME_EPI_BF16_GELU: return YUME_GO(YUME_EPI_BF16_GELU);
        case YUME_EPI_BF16_GELU_ERF: return YUME_GO(YUME_EPI_BF16_GELU_ERF);
        case YUME_EPI_F32: return YUME_GO(YUME_EPI_F32);
        case YUME_EPI_RESID:
            YUME_REQUIRE(gate == nullptr || (gate_stride % 4) == 0, "gemm_bf16: gate_stride must be a multiple of 4");
            return YUME_GO(YUME_EPI_RESID);
        case YUME_EPI_BF16_GEGLU:
            YUME_REQUIRE((N % 8) == 0 && bias == nullptr, "gemm_bf16: GEGLU needs N %% 8 == 0 and no bias");
            return launch256<YUME_EPI_BF16_GEGLU>(p, al, e, st, "gemm_bf16");     // 256x256 kernel only (vector epilogue)
        case YUME_EPI_BF16_SPLITT:
            YUME_REQUIRE(outT != nullptr && n_split >= 0 && (n_split % BN) == 0 && ldt >= M && (ldt % 4) == 0,
                         "gemm_bf16: SPLITT needs outT, n_split %% 128 == 0, ldt >= M and ldt %% 4 == 0");
            YUME_REQUIRE(((uintptr_t)outT % 16) == 0, "gemm_bf16: outT must be 16-byte aligned");
            return YUME_GO(YUME_EPI_BF16_SPLITT);
        default:
            yume_set_error("gemm_bf16: unknown epilogue %d", epi);
            return YUME_EINVAL;
    }
}

// `batch` independent small GEMMs of one shape (the per-head score / value products of the T5 encoder's attention) in ONE
// launch of the 128x128 kernel (gridDim.y = batch). Strides are in ELEMENTS of the respective tensor.
extern "C" int yume_gemm_bf16_batched(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw, int64_t strideW,
                                      int64_t M, int64_t N, int64_t K, int epi, void* out, int64_t ldo, int64_t strideO,
                                      int64_t batch, int variant, void* stream) {
    (void)variant;
    YUME_REQUIRE(A && W && out, "gemm_bf16_batched: NULL pointer");
    YUME_REQUIRE(epi == YUME_EPI_BF16 || epi == YUME_EPI_F32, "gemm_bf16_batched: only the plain bf16 / fp32 epilogues");
    YUME_REQUIRE(batch >= 1 && batch <= 65535, "gemm_bf16_batched: batch %lld", (long long)batch);
    YUME_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "gemm_bf16_batched: bad problem size");
    YUME_REQUIRE((K % BK) == 0 && (N % 4) == 0, "gemm_bf16_batched: K must be a multiple of %d and N of 4", BK);
    YUME_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0 && (ldo % 4) == 0, "gemm_bf16_batched: lda/ldw must be multiples of 8, ldo of 4");
    YUME_REQUIRE((strideA % 8) == 0 && (strideW % 8) == 0 && (strideO % 8) == 0, "gemm_bf16_batched: strides must keep 16-byte alignment");
    YUME_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_bf16_batched: pointers must be 16-byte aligned");
    Problem p;
    p.W = (const unsigned short*)W; p.ldw = ldw;
    p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.tiles_m = (int)((M + BM - 1) / BM);
    p.tiles_n = (int)((N + BN - 1) / BN);
    p.group_m = 8;
    p.bsW = strideW;
    p.bsO = strideO * (epi == YUME_EPI_F32 ? 4 : 2);
    PlainA al;
    al.A = (const unsigned short*)A; al.lda = lda; al.M = (int)M; al.bsA = strideA;
    Epilogue e = {};
    e.out = out; e.ldo = ldo;
    hipStream_t st = (hipStream_t)stream;
    if (epi == YUME_EPI_F32) return launch<YUME_EPI_F32>(p, al, e, st, "gemm_bf16_batched", (int)batch);
    return launch<YUME_EPI_BF16>(p, al, e, st, "gemm_bf16_batched", (int)batch);
}

// ---- split-K for small-M, weight-streaming GEMMs (text / vision encoders: M <= 512 tokens against N x K weights of tens of MB)
// With M <= 512 the 128x128 tiling has at most 4 * N/128 workgroups (32..160 for the encoder shapes): a fraction of the 256
// CUs streams the weights and the GEMM runs at ~1 TB/s. Here K is cut into `splits` slices computed side by side (the batched
// launch, gridDim.y = splits, fp32 partials in the caller's workspace [splits, M, N]) and a second kernel sums the slices in a
// fixed order and applies bias + epilogue. HBM-bound: algorithmic bytes = N*K*2 (the weights, read once).
namespace {
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int64_t MN, int N,
                                                            const float* __restrict__ bias, void* __restrict__ out, int64_t ldo) {
    const int64_t nq = MN >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = *reinterpret_cast<const f32x4*>(ws + 4 * i);
        for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(ws + s * MN + 4 * i);
        const int64_t m = (4 * i) / N;
        const int n = (int)((4 * i) - m * N);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (EPI == YUME_EPI_F32) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + m * ldo + n) = v;
        } else if (EPI == YUME_EPI_RESID) {
            float* xo = reinterpret_cast<float*>(out) + m * ldo + n;
            *reinterpret_cast<f32x4*>(xo) = *reinterpret_cast<const f32x4*>(xo) + v;
        } else if (EPI == YUME_EPI_BF16_GEGLU) {
            const unsigned o = pack_bf16x2(v[1] * gelu_tanh(v[0]), v[3] * gelu_tanh(v[2]));
            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(out) + m * ldo + (n >> 1)) = o;
        } else {
            if (EPI == YUME_EPI_BF16_GELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = gelu_tanh(v[q]);
            }
            if (EPI == YUME_EPI_BF16_GELU_ERF) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = 0.5f * v[q] * (1.0f + erff(v[q] * 0.7071067811865476f));
            }
            u32x2 o;
            o[0] = pack_bf16x2(v[0], v[1]);
            o[1] = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(out) + m * ldo + n) = o;
        }
    }
}
}  // namespace

extern "C" int64_t yume_gemm_splitk_workspace_bytes(int64_t M, int64_t N, int splits) { return (int64_t)splits * M * N * 4; }

extern "C" int yume_gemm_bf16_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int64_t M, int64_t N,
                                     int64_t K, int epi, void* out, int64_t ldo, int splits, void* workspace, void* stream) {
    YUME_REQUIRE(A && W && out && workspace, "gemm_bf16_splitk: NULL pointer");
    YUME_REQUIRE(splits >= 1 && splits <= 64 && (K % (splits * BK)) == 0, "gemm_bf16_splitk: K=%lld must be a multiple of splits*%d (splits=%d)",
                 (long long)K, BK, splits);
    YUME_REQUIRE(M > 0 && N > 0 && (N % 8) == 0 && (ldo % 4) == 0, "gemm_bf16_splitk: N must be a multiple of 8, ldo of 4");
    YUME_REQUIRE(((uintptr_t)workspace % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_bf16_splitk: pointers must be 16-byte aligned");
    YUME_REQUIRE((M * N) % 4 == 0, "gemm_bf16_splitk: M*N must be a multiple of 4");
    const int64_t Ks = K / splits;
    int rc = yume_gemm_bf16_batched(A, lda, Ks, W, ldw, Ks, M, N, Ks, YUME_EPI_F32, workspace, N, M * N, splits, 1, stream);
    if (rc != YUME_OK) return rc;
    const int64_t MN = M * N;
    const unsigned grid = (unsigned)((MN / 4 + 255) / 256 > 16384 ? 16384 : (MN / 4 + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    const float* ws = (const float*)workspace;
#define YUME_RED(E) hipLaunchKernelGGL((splitk_reduce_kernel<E>), dim3(grid), dim3(256), 0, st, ws, splits, MN, (int)N, bias, out, ldo)
    switch (epi) {
        case YUME_EPI_F32: YUME_RED(YUME_EPI_F32); break;
        case YUME_EPI_RESID: YUME_RED(YUME_EPI_RESID); break;
        case YUME_EPI_BF16: YUME_RED(YUME_EPI_BF16); break;
        case YUME_EPI_BF16_GELU: YUME_RED(YUME_EPI_BF16_GELU); break;
        case YUME_EPI_BF16_GELU_ERF: YUME_RED(YUME_EPI_BF16_GELU_ERF); break;
        case YUME_EPI_BF16_GEGLU: YUME_RED(YUME_EPI_BF16_GEGLU); break;
        default:
            yume_set_error("gemm_bf16_splitk: unsupported epilogue %d", epi);
            return YUME_EINVAL;
    }
    YUME_CHECK_LAUNCH("gemm_bf16_splitk");
    return YUME_OK;
}
