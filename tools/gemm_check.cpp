// gemm_check — standalone (no torch) correctness + timing harness for yume_gemm_bf16.
//   build:  hipcc -O2 --offload-arch=gfx950 tools/gemm_check.cpp -o tools/gemm_check -ldl
//   run:    [YUME_GEMM_MODE=m] tools/gemm_check [--lib path] [--timing]
// Every epilogue is checked on small / ragged shapes against an fp64 host product (test infrastructure, like oracle/); the
// DiT shapes are checked on sampled output elements (fp64 dot products) and timed with HIP events.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/yume_hip.h"

typedef int (*gemm_fn)(const void*, int64_t, const void*, int64_t, const float*, int64_t, int64_t, int64_t, int, void*, int64_t, const float*, int64_t,
                       const int32_t*, void*, int64_t, int64_t, int, void*);
typedef const char* (*err_fn)();
typedef int (*gemm_ws_fn)(const void*, int64_t, const void*, int64_t, const float*, int64_t, int64_t, int64_t, int, void*, int64_t, const float*, int64_t,
                          const int32_t*, void*, int64_t, int64_t, int, void*, int64_t, void*);
typedef int64_t (*ws_bytes_fn)();
static gemm_fn p_gemm;
static gemm_ws_fn p_gemm_ws;      // --sk: the stream-K entry point with a zero-initialised scratch
static void* g_ws;
static int64_t g_ws_bytes;
static err_fn p_err;

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint64_t rs = 0x1234567ull;
static float rnd() { float s = 0; for (int i = 0; i < 4; ++i) { rs = rs * 6364136223846793005ull + 1442695040888963407ull; s += (float)((rs >> 33) & 0xffffff) / 16777216.0f - 0.5f; } return s * 1.7320508f; }
static float gelu_tanh(float x) { const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x); return 0.5f * x * (1.0f + tanhf(u)); }

struct Case { int64_t M, N, K; int epi; int R; };   // R: gate rows (RESID), 0 = no gate

static int run_case(const Case& c, const std::vector<int>& variants, int reps, bool full_check, bool timing) {
    int total_bad = 0;
    const int64_t M = c.M, N = c.N, K = c.K;
    std::vector<uint16_t> a(M * K), w(N * K);
    const float ws = 1.0f / sqrtf((float)K);
    for (auto& x : a) x = f2bf(rnd());
    for (auto& x : w) x = f2bf(rnd() * ws);
    std::vector<float> bias(N), gate((size_t)(c.R > 0 ? c.R : 1) * N), x0(M * N);
    for (auto& x : bias) x = rnd() * 0.1f;
    for (auto& x : gate) x = rnd();
    for (auto& x : x0) x = rnd();
    std::vector<int32_t> ridx(M);
    // gate rows: interleaved for the small cases (every wave sees several), in segments as the tokens' timesteps are for the big ones (a wave's
    // 128 rows share one row except across the segment boundaries, which fall inside tiles)
    for (int64_t m = 0; m < M; ++m) ridx[m] = c.R > 1 ? (M >= 2048 ? (int32_t)(m * c.R / M) : (int32_t)((m * 7) % c.R)) : 0;
    const int64_t nsplit = c.epi == YUME_EPI_BF16_SPLITT ? (N / 3 / 128) * 128 * 2 : 0;      // q|k row-major, v transposed
    const int64_t ldt = (M + 7) / 8 * 8;
    void *da, *dw, *dout, *dt = nullptr; float *dbias, *dgate; int32_t* dr;
    const size_t osz = (c.epi == YUME_EPI_F32 || c.epi == YUME_EPI_RESID) ? 4 : 2;
    HC(hipMalloc(&da, a.size() * 2)); HC(hipMalloc(&dw, w.size() * 2)); HC(hipMalloc(&dout, M * N * osz));
    HC(hipMalloc(&dbias, N * 4)); HC(hipMalloc(&dgate, gate.size() * 4)); HC(hipMalloc(&dr, M * 4));
    HC(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice)); HC(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    HC(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice)); HC(hipMemcpy(dgate, gate.data(), gate.size() * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(dr, ridx.data(), M * 4, hipMemcpyHostToDevice));
    if (c.epi == YUME_EPI_BF16_SPLITT) { HC(hipMalloc(&dt, (N - nsplit) * ldt * 2)); HC(hipMemset(dt, 0, (N - nsplit) * ldt * 2)); }
    for (int rep = 0; rep < reps; ++rep)
    for (int variant : variants) {
    const char* tag = variant == 3 ? (full_check ? "small/w4 " : "big/w4  ") : variant == 2 ? (full_check ? "small/256" : "big/256 ") : (full_check ? "small/auto" : "big/auto");
    if (c.epi == YUME_EPI_RESID) HC(hipMemcpy(dout, x0.data(), M * N * 4, hipMemcpyHostToDevice));
    else HC(hipMemset(dout, 0xff, M * N * osz));
    if (dt) HC(hipMemset(dt, 0, (N - nsplit) * ldt * 2));
    auto call = [&]() {
        if (g_ws) return p_gemm_ws(da, K, dw, K, dbias, M, N, K, c.epi, dout, N, c.R > 0 ? dgate : nullptr, N, c.R > 1 ? dr : nullptr, dt, ldt, nsplit, variant, g_ws, g_ws_bytes, nullptr);
        return p_gemm(da, K, dw, K, dbias, M, N, K, c.epi, dout, N, c.R > 0 ? dgate : nullptr, N, c.R > 1 ? dr : nullptr, dt, ldt, nsplit, variant, nullptr);
    };
    int rc = call();
    if (rc) { printf("%s M=%lld N=%lld K=%lld epi=%d: rc=%d %s\n", tag, (long long)M, (long long)N, (long long)K, c.epi, rc, p_err()); return 1; }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("device error %s\n", hipGetErrorString(e)); exit(3); }
    std::vector<uint8_t> out(M * N * osz);
    HC(hipMemcpy(out.data(), dout, out.size(), hipMemcpyDeviceToHost));
    std::vector<uint16_t> outT;
    if (dt) { outT.resize((N - nsplit) * ldt); HC(hipMemcpy(outT.data(), dt, outT.size() * 2, hipMemcpyDeviceToHost)); }
    // ---- check
    double maxerr = 0, maxref = 0; int bad = 0;
    auto check = [&](int64_t m, int64_t n) {
        double acc = 0;
        for (int64_t k = 0; k < K; ++k) acc += (double)bf2f(a[m * K + k]) * bf2f(w[n * K + k]);
        acc += bias[n];
        double want, got, tol;
        if (c.epi == YUME_EPI_RESID) {
            want = x0[m * N + n] + acc * (c.R > 0 ? gate[(size_t)ridx[m] * N + n] : 1.0);
            got = ((float*)out.data())[m * N + n]; tol = 2e-3;
        } else if (c.epi == YUME_EPI_F32) {
            want = acc; got = ((float*)out.data())[m * N + n]; tol = 2e-3;
        } else {
            want = c.epi == YUME_EPI_BF16_GELU ? gelu_tanh((float)acc) : acc;
            if (c.epi == YUME_EPI_BF16_SPLITT && n >= nsplit) got = bf2f(outT[(n - nsplit) * ldt + m]);
            else got = bf2f(((uint16_t*)out.data())[m * N + n]);
            tol = 1e-2 * (fabs(want) > 1 ? fabs(want) : 1) + 2e-3;
        }
        const double err = fabs(got - want);
        if (!(err <= tol)) ++bad;
        maxerr = err > maxerr ? err : maxerr; maxref = fabs(want) > maxref ? fabs(want) : maxref;
    };
    if (full_check) {
        for (int64_t m = 0; m < M; ++m) for (int64_t n = 0; n < N; ++n) check(m, n);
        // nothing may be written outside the matrix: the K-major image's pad columns [M, ldt) stay as memset
        if (dt) for (int64_t n = 0; n < N - nsplit; ++n) for (int64_t m = M; m < ldt; ++m) if (outT[n * ldt + m] != 0) ++bad;
    } else if (rep == 0) {
        for (int i = 0; i < 4096; ++i) {
            rs = rs * 6364136223846793005ull + 1442695040888963407ull;
            int64_t m = (rs >> 20) % M, n = (rs >> 40) % N;
            if (i < 64) { m = (i & 1) ? M - 1 - (i >> 1) % 16 : (i >> 1) % 16; }          // first and last rows
            else if (i < 128) { n = (i & 1) ? N - 1 - (i >> 1) % 16 : (i >> 1) % 16; }
            check(m, n);
        }
    }
    float ms = 0;
    if (timing) {
        hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) call();
        HC(hipEventRecord(e0, nullptr));
        const int it = 20;
        for (int i = 0; i < it; ++i) call();
        HC(hipEventRecord(e1, nullptr)); HC(hipEventSynchronize(e1));
        HC(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    }
    printf("%s M=%lld N=%lld K=%lld epi=%d R=%d variant=%d  maxerr=%.3e (ref max %.2f) bad=%d %s", tag, (long long)M, (long long)N, (long long)K, c.epi, c.R,
           variant, maxerr, maxref, bad, bad ? "FAIL" : "ok");
    if (timing) printf("   %.4f ms  %.0f TFLOP/s", ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
    printf("\n");
    fflush(stdout);
    total_bad += bad;
    }
    hipFree(da); hipFree(dw); hipFree(dout); hipFree(dbias); hipFree(dgate); hipFree(dr); if (dt) hipFree(dt);
    return total_bad ? 1 : 0;
}

int main(int argc, char** argv) {
    const char* lib = "yume_amd/lib/libyume_hip.so";
    bool timing_only = false, quick = false, expm = false;
    std::vector<int> variants = {2, 3};        // 2 = 8-wave 256x256 kernel, 3 = one-wave-per-SIMD 256x256 kernel (gemm_w4.hpp), 0 = automatic
    int reps = 2;
    bool use_sk = false;              // --sk: yume_gemm_bf16_ws with a scratch (stream-K tail of the one-wave-per-SIMD kernel)
    const char* trace = nullptr;      // --trace file (with --one): dump the per-workgroup time stamps of an experiment build (csrc/trace.hpp)
    long long one[5] = {0, 0, 0, 0, 0};
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--lib")) lib = argv[++i];
        else if (!strcmp(argv[i], "--timing")) timing_only = true;
        else if (!strcmp(argv[i], "--quick")) { timing_only = true; quick = true; }     // the 5B shapes + 8192^3, the listed variants interleaved
        else if (!strcmp(argv[i], "--exp")) { timing_only = true; expm = true; }         // experiment builds (bf16 epilogue only): the block's shapes with a plain epilogue
        else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--trace")) trace = argv[++i];
        else if (!strcmp(argv[i], "--sk")) use_sk = true;
        else if (!strcmp(argv[i], "--one")) { for (int j = 0; j < 5; ++j) one[j] = atoll(argv[++i]); timing_only = true; }      // M N K epi row_idx_tables
        else if (!strcmp(argv[i], "--variants")) {
            variants.clear();
            for (char* t = strtok(argv[++i], ","); t; t = strtok(nullptr, ",")) variants.push_back(atoi(t));
        }
    }
    void* hnd = dlopen(lib, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
    if (!hnd) { printf("dlopen %s: %s\n", lib, dlerror()); return 2; }
    p_gemm = (gemm_fn)dlsym(hnd, "yume_gemm_bf16"); p_err = (err_fn)dlsym(hnd, "yume_last_error");
    if (use_sk) {
        p_gemm_ws = (gemm_ws_fn)dlsym(hnd, "yume_gemm_bf16_ws");
        ws_bytes_fn wb = (ws_bytes_fn)dlsym(hnd, "yume_gemm_workspace_bytes");
        if (!p_gemm_ws || !wb) { printf("--sk: the library has no yume_gemm_bf16_ws\n"); return 2; }
        g_ws_bytes = wb();
        HC(hipMalloc(&g_ws, g_ws_bytes));
        HC(hipMemset(g_ws, 0, g_ws_bytes));
    }
    const char* mode = getenv("YUME_GEMM_MODE");
    printf("library %s  YUME_GEMM_MODE=%s\n", lib, mode ? mode : "(default)");
    int fails = 0;
    if (!timing_only) {
        const Case small[] = {{256, 256, 64, YUME_EPI_F32, 0}, {256, 256, 128, YUME_EPI_F32, 0}, {512, 512, 192, YUME_EPI_BF16, 0}, {300, 512, 256, YUME_EPI_F32, 0},
                              {257, 768, 320, YUME_EPI_BF16_GELU, 0}, {1000, 512, 64, YUME_EPI_RESID, 0}, {777, 1024, 448, YUME_EPI_RESID, 3},
                              {520, 768, 256, YUME_EPI_BF16_SPLITT, 0}, {1024, 1536, 512, YUME_EPI_BF16_SPLITT, 0}, {256, 512, 1024, YUME_EPI_RESID, 1},
                              {2048, 256, 3072, YUME_EPI_F32, 0}, {511, 2048, 576, YUME_EPI_BF16, 0},
                              // K tiles 2 .. 7 (every prologue / tail shape of the two-tiles-ahead pipeline), ragged M and N edges, every epilogue
                              {512, 512, 128, YUME_EPI_BF16, 0}, {512, 256, 256, YUME_EPI_RESID, 2}, {260, 516, 320, YUME_EPI_BF16, 0},
                              {1000, 768, 384, YUME_EPI_BF16_GELU, 0}, {999, 1000, 448, YUME_EPI_F32, 0}, {770, 2304, 192, YUME_EPI_BF16_SPLITT, 0},
                              {1290, 1536, 640, YUME_EPI_BF16_SPLITT, 0}, {513, 260, 128, YUME_EPI_RESID, 0}, {768, 1024, 1024, YUME_EPI_RESID, 5}};
        for (auto& c : small) fails += run_case(c, variants, 1, true, false);
    }
    const Case big[] = {{9460, 9216, 3072, YUME_EPI_BF16_SPLITT, 0}, {9460, 3072, 3072, YUME_EPI_RESID, 2}, {9460, 3072, 3072, YUME_EPI_RESID, 0},
                        {9460, 3072, 3072, YUME_EPI_BF16, 0}, {9460, 14336, 3072, YUME_EPI_BF16_GELU, 0}, {9460, 3072, 14336, YUME_EPI_RESID, 2},
                        {8192, 8192, 8192, YUME_EPI_BF16, 0}, {27810, 5120, 5120, YUME_EPI_RESID, 1}, {27810, 13824, 5120, YUME_EPI_BF16_GELU, 0}};
    if (one[0]) {
        const Case c = {(int)one[0], (int)one[1], (int)one[2], (int)one[3], (int)one[4]};
        fails += run_case(c, variants, reps, false, true);
        if (trace) {
            typedef int (*trace_fn)(void*, long long);
            trace_fn rd = (trace_fn)dlsym(hnd, "yume_debug_trace_read");
            if (!rd) { printf("--trace: %s is not a -DYUME_TRACE build\n", lib); return 2; }
            std::vector<unsigned long long> t(32768 * 8);
            hipDeviceSynchronize();
            if (rd(t.data(), (long long)t.size() * 8)) { printf("trace read failed\n"); return 2; }
            FILE* f = fopen(trace, "wb"); fwrite(t.data(), 8, t.size(), f); fclose(f);
            printf("trace written to %s\n", trace);
        }
        return fails ? 1 : 0;
    }
    if (expm) {
        const Case ex[] = {{9460, 3072, 3072, YUME_EPI_BF16, 0}, {9460, 14336, 3072, YUME_EPI_BF16, 0}, {9460, 3072, 14336, YUME_EPI_BF16, 0}, {8192, 8192, 8192, YUME_EPI_BF16, 0}};
        for (auto& c : ex) fails += run_case(c, variants, reps, false, true);
        printf("gemm_check: %d failure(s)\n", fails);
        return fails ? 1 : 0;
    }
    int nb = 0;
    for (auto& c : big) {
        if (quick && nb++ >= 7) break;
        std::vector<int> vs = variants;
        if (!quick) vs.push_back(0);
        fails += run_case(c, vs, quick ? reps : 1, false, true);
    }
    printf("gemm_check: %d failure(s)\n", fails);
    return fails ? 1 : 0;
}
