// gemm_bf16.hip — bf16 MFMA GEMM  acc[m,n] = sum_k A[m,k] * W[n,k]  with fused epilogues (gfx950).
//
// Tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves in a 2x2 grid, each wave 64x64 = 4x4
// v_mfma_f32_16x16x32_bf16 accumulators. Both operands are K-contiguous (A row-major activations,
// W in nn.Linear [out,in] layout) so the A and B tiles are staged identically:
//   HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane), double buffered, one barrier per K tile.
//   The LDS image is lane-linear; the bank-conflict swizzle is applied on the SOURCE address
//   (16-byte chunk c of row r is fetched from chunk c ^ (r & 7)) and undone on the ds_read_b128.
// Epilogue: accumulators are restaged through LDS per wave so every global access is a full
// 8/16-byte vector along the contiguous dimension, and bias / GELU / gate*y+residual / transposed
// (K-major V^T) stores are fused there.
// Workgroup order: bijective XCD remap (block b runs on XCD b%8) + grouped (8 M-tiles) traversal so
// the blocks resident on one XCD share A row-panels and W column-panels in that XCD's L2.
// Roofline: MFMA (bf16 dense 2.5 PFLOP/s); algorithmic work 2*M*N*K flop per launch.
#include "common.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHR = 256;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;       // double buffered = 64 KiB

struct GemmArgs {
    const unsigned short* A; int64_t lda;
    const unsigned short* W; int64_t ldw;
    const float* bias;
    int M, N, K;
    void* out; int64_t ldo;
    const float* gate; int64_t gate_stride; const int32_t* row_idx;
    unsigned short* outT; int64_t ldt; int n_split;
    int tiles_m, tiles_n;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

__device__ __forceinline__ void stage_tile(const unsigned short* __restrict__ src, int64_t ld, int row0, int rows_max,
                                           int k0, char* lds_tile, int tid, int wave) {
    // 128 rows x 128 B. round rr covers rows rr*32 .. +31: thread t -> (row = rr*32 + t/8, lds chunk = t%8)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = rr * 32 + (tid >> 3);
        const int c = tid & 7;
        const int gc = c ^ (r & 7);
        int gr = row0 + r;
        gr = gr < rows_max ? gr : rows_max - 1;
        const unsigned short* g = src + (int64_t)gr * ld + k0 + gc * 8;
        char* l = lds_tile + rr * 4096 + wave * 1024;  // wave-uniform base; hardware adds lane*16
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)g, (lds_void*)l, 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8_t lds_frag(const char* tile, int row, int chunk) {
    const char* p = tile + row * 128 + ((chunk ^ (row & 7)) << 4);
    return *reinterpret_cast<const bf16x8_t*>(p);
}

template <int EPI>
__global__ __launch_bounds__(NTHR, 2) void gemm128_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

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

    const int nk = p.K / BK;
    stage_tile(p.A, p.lda, m0, p.M, 0, smem, tid, wave);
    stage_tile(p.W, p.ldw, n0, p.N, 0, smem + TILE_BYTES, tid, wave);

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed (own loads) ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... for every wave, and every wave is done reading the other buffer
        __syncthreads();
        if (kt + 1 < nk) {
            char* nb = smem + (cur ^ 1) * STAGE_BYTES;
            stage_tile(p.A, p.lda, m0, p.M, (kt + 1) * BK, nb, tid, wave);
            stage_tile(p.W, p.ldw, n0, p.N, (kt + 1) * BK, nb + TILE_BYTES, tid, wave);
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

    // ---- epilogue: per-wave 64x64 fp32 restage (16 KiB per wave) ----
    float* ep = reinterpret_cast<float*>(smem) + wave * (64 * 64);
    const int wm0 = m0 + wm * 64, wn0 = n0 + wn * 64;
    const bool transposed = (EPI == YUME_EPI_BF16_SPLITT) && (n0 >= p.n_split);
    if (!transposed) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ep[(i * 16 + 4 * (lane >> 4) + r) * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
    } else {
        // [n][m] image, 4-float granule g of row n stored at granule g ^ (n & 15)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = j * 16 + (lane & 15);
                const int g = i * 4 + (lane >> 4);
                *reinterpret_cast<f32x4*>(ep + n * 64 + ((g ^ (n & 15)) << 2)) = acc[i][j];
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own LDS writes done (region is private to the wave)
    __builtin_amdgcn_wave_barrier();

    const int sub = lane >> 4;         // row within a pass of 4
    const int c4 = (lane & 15) << 2;   // first of 4 contiguous columns
    if (!transposed) {
        const int n = wn0 + c4;
        f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll 4
        for (int ps = 0; ps < 16; ++ps) {
            const int rl = ps * 4 + sub;
            const int m = wm0 + rl;
            if (m >= p.M || n >= p.N) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(ep + rl * 64 + c4);
            v += bias4;
            if (EPI == YUME_EPI_BF16 || EPI == YUME_EPI_BF16_GELU || EPI == YUME_EPI_BF16_SPLITT ||
                EPI == YUME_EPI_BF16_GELU_ERF) {
                if (EPI == YUME_EPI_BF16_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
                }
                if (EPI == YUME_EPI_BF16_GELU_ERF) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.7071067811865476f));
                }
                u32x2 o;
                o[0] = pack_bf16x2(v[0], v[1]);
                o[1] = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(p.out) + (int64_t)m * p.ldo + n) = o;
            } else if (EPI == YUME_EPI_F32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n) = v;
            } else {  // YUME_EPI_RESID
                float* xo = reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n;
                f32x4 x = *reinterpret_cast<const f32x4*>(xo);
                if (p.gate) {
                    const int64_t row = p.row_idx ? (int64_t)p.row_idx[m] : 0;
                    const f32x4 g = *reinterpret_cast<const f32x4*>(p.gate + row * p.gate_stride + n);
                    x += v * g;
                } else {
                    x += v;
                }
                *reinterpret_cast<f32x4*>(xo) = x;
            }
        }
    } else {
        // rows of the image are output features n, columns are tokens m
        const int m = wm0 + c4;
#pragma unroll 4
        for (int ps = 0; ps < 16; ++ps) {
            const int nl = ps * 4 + sub;
            const int n = wn0 + nl;
            if (n >= p.N || m >= p.M) continue;
            const int g = lane & 15;
            f32x4 v = *reinterpret_cast<const f32x4*>(ep + nl * 64 + ((g ^ (nl & 15)) << 2));
            const float bn = p.bias ? p.bias[n] : 0.f;
            unsigned short* dst = p.outT + (int64_t)(n - p.n_split) * p.ldt + m;
            if (m + 3 < p.M) {
                u32x2 o;
                o[0] = pack_bf16x2(v[0] + bn, v[1] + bn);
                o[1] = pack_bf16x2(v[2] + bn, v[3] + bn);
                *reinterpret_cast<u32x2*>(dst) = o;
            } else {
                for (int e = 0; e < 4 && m + e < p.M; ++e) dst[e] = f32_to_bf16(v[e] + bn);
            }
        }
    }
}

template <int EPI>
int launch(const GemmArgs& a, hipStream_t st) {
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(NTHR);
    hipLaunchKernelGGL(gemm128_kernel<EPI>, grid, block, 0, st, a);
    YUME_CHECK_LAUNCH("gemm_bf16");
    return YUME_OK;
}

}  // namespace

extern "C" int yume_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int64_t M,
                              int64_t N, int64_t K, int epi, void* out, int64_t ldo, const float* gate,
                              int64_t gate_stride, const int32_t* row_idx, void* outT, int64_t ldt, int64_t n_split,
                              int variant, void* stream) {
    (void)variant;
    YUME_REQUIRE(A && W && out, "gemm_bf16: NULL pointer");
    YUME_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_bf16: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    YUME_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "gemm_bf16: dimension too large");
    YUME_REQUIRE((K % BK) == 0, "gemm_bf16: K=%lld must be a multiple of %d", (long long)K, BK);
    YUME_REQUIRE((N % 4) == 0, "gemm_bf16: N=%lld must be a multiple of 4", (long long)N);
    YUME_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0, "gemm_bf16: lda/ldw must be multiples of 8 (16-byte rows)");
    YUME_REQUIRE((ldo % 4) == 0, "gemm_bf16: ldo must be a multiple of 4");
    YUME_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_bf16: pointers must be 16-byte aligned");
    GemmArgs a;
    a.A = (const unsigned short*)A; a.lda = lda;
    a.W = (const unsigned short*)W; a.ldw = ldw;
    a.bias = bias;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.out = out; a.ldo = ldo;
    a.gate = gate; a.gate_stride = gate_stride; a.row_idx = row_idx;
    a.outT = (unsigned short*)outT; a.ldt = ldt; a.n_split = (int)n_split;
    a.tiles_m = (int)((M + BM - 1) / BM);
    a.tiles_n = (int)((N + BN - 1) / BN);
    hipStream_t st = (hipStream_t)stream;
    switch (epi) {
        case YUME_EPI_BF16: return launch<YUME_EPI_BF16>(a, st);
        case YUME_EPI_BF16_GELU: return launch<YUME_EPI_BF16_GELU>(a, st);
        case YUME_EPI_BF16_GELU_ERF: return launch<YUME_EPI_BF16_GELU_ERF>(a, st);
        case YUME_EPI_F32: return launch<YUME_EPI_F32>(a, st);
        case YUME_EPI_RESID:
            YUME_REQUIRE(gate == nullptr || (gate_stride % 4) == 0, "gemm_bf16: gate_stride must be a multiple of 4");
            return launch<YUME_EPI_RESID>(a, st);
        case YUME_EPI_BF16_SPLITT:
            YUME_REQUIRE(outT != nullptr && n_split >= 0 && (n_split % BN) == 0 && ldt >= M && (ldt % 4) == 0,
                         "gemm_bf16: SPLITT needs outT, n_split %% 128 == 0, ldt >= M and ldt %% 4 == 0");
            YUME_REQUIRE(((uintptr_t)outT % 16) == 0, "gemm_bf16: outT must be 16-byte aligned");
            return launch<YUME_EPI_BF16_SPLITT>(a, st);
        default:
            yume_set_error("gemm_bf16: unknown epilogue %d", epi);
            return YUME_EINVAL;
    }
}
