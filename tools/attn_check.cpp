// attn_check — standalone (no torch) correctness + timing harness for yume_attn_fwd variants.
//   build:  hipcc -O2 --offload-arch=gfx950 tools/attn_check.cpp -o tools/attn_check -ldl      (NOT linked against the library: a second copy loaded first would capture the --lib build's calls)
//   run:    tools/attn_check [variants...]        (default variants: 7 263 256 776 768 4 2 0; v + 256 = v | YUME_ATTN_Q_PRESCALED: the harness
//           hands the kernel q' = bf16(q * scale * log2 e) and the references take exp2(q' . k); v + 512 = v | YUME_ATTN_KV_PADDED: the
//           harness hands it K / V^T buffers padded to whole 64-key tiles — NaN rows behind K, finite junk behind V^T's columns; 776 =
//           8 | 256 | 512 is the persistent kernel, which is also held BIT-IDENTICAL to variant 263 on launches without scratch)
// Small shapes are checked against an fp64 exact-softmax reference computed on the host (test infrastructure, like oracle/);
// large shapes are checked against variant 2 (itself checked on the small shapes) and timed with HIP events.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <dlfcn.h>
#include "../include/yume_hip.h"

// the library under test is dlopen()ed so that timing-experiment builds (other .so files) can be compared by the same binary
typedef int (*attn_fn)(const void*, int64_t, const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t, int64_t, float, int, int, void*, int64_t, void*);
typedef int64_t (*ws_fn)(int64_t, int64_t, int64_t);
typedef const char* (*err_fn)();
typedef int64_t (*cwb_fn)();
typedef int (*cwi_fn)(void*, int64_t, void*);
static attn_fn p_attn; static ws_fn p_ws; static err_fn p_err;
#define yume_attn_fwd_ws p_attn
#define yume_attn_workspace_bytes p_ws
#define yume_last_error p_err

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint16_t f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float rnd() {   // ~N(0,1): sum of 4 uniforms
    float s = 0;
    for (int i = 0; i < 4; ++i) {
        rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
        s += (float)((rng_state >> 33) & 0xffffff) / 16777216.0f - 0.5f;
    }
    return s * 1.7320508f;
}

struct Prob {
    int64_t Lq, Lk, H, ldq, ldk, ldvt, ldo;
    std::vector<uint16_t> q, qpre, k, vt, o0;
    uint16_t *dq, *dqpre, *dk, *dvt, *d_o;
    uint16_t *dkp, *dvtp; int64_t ldvtp;      // YUME_ATTN_KV_PADDED images: K with NaN rows up to a whole tile, V^T with finite junk columns
    void* ws; int64_t wsb;
};

static const double kScale = 0.08838834764831845, kLog2e = 1.4426950408889634;
// spike: 0 none, 1 one key far above the rest late in the sequence (rescale / redo path), 2 scores far outside +-100 in the log2 domain for
// some queries (huge positive and huge negative rows: the base-free pieces must hand the workgroup to the robust ones)
static void make(Prob& p, int64_t Lq, int64_t Lk, int64_t H, int spike) {
    p.Lq = Lq; p.Lk = Lk; p.H = H;
    p.ldq = H * 128; p.ldk = H * 128; p.ldo = H * 128;
    p.ldvt = (Lk + 7) / 8 * 8;
    p.q.resize(Lq * p.ldq); p.k.resize(Lk * p.ldk); p.vt.assign(H * 128 * p.ldvt, 0x7fc0 /* NaN in the padding: must never leak */);
    p.o0.resize(Lq * p.ldo);
    for (auto& x : p.q) x = f2bf(rnd());
    for (auto& x : p.k) x = f2bf(rnd());
    for (int64_t r = 0; r < H * 128; ++r)
        for (int64_t c = 0; c < Lk; ++c) p.vt[r * p.ldvt + c] = f2bf(rnd());
    for (auto& x : p.o0) x = f2bf(rnd());
    if (spike) {
        // one key far above the rest for some queries, late in the sequence: forces the rescale / redo path in a late tile
        int64_t key = Lk - 1 - (Lk > 70 ? 37 : 0);
        for (int64_t h = 0; h < H; ++h)
            for (int d = 0; d < 128; ++d) {
                float qv = bf2f(p.q[(Lq / 2) * p.ldq + h * 128 + d]);
                p.k[key * p.ldk + h * 128 + d] = f2bf(qv * 6.0f);
            }
    }
    if (spike == 2) {
        // query row 3: key 5 = 40 x that query -> a score of several hundred (log2 domain); query row Lq - 2: every key carries a constant
        // +8 in feature 0 (a uniform shift for the other queries) and this query -400 there -> every one of its scores is about -400
        const int64_t q1 = 3 < Lq ? 3 : 0, q2 = Lq - 2 >= 0 ? Lq - 2 : 0, k1 = 5 < Lk ? 5 : 0;
        for (int64_t h = 0; h < H; ++h) {
            for (int64_t j = 0; j < Lk; ++j) p.k[j * p.ldk + h * 128] = f2bf(8.0f);
            if (q2 != q1) p.q[q2 * p.ldq + h * 128] = f2bf(-400.0f);
            for (int d = 0; d < 128; ++d) p.k[k1 * p.ldk + h * 128 + d] = f2bf(bf2f(p.q[q1 * p.ldq + h * 128 + d]) * 40.0f);
        }
    }
    p.qpre.resize(p.q.size());
    for (size_t i = 0; i < p.q.size(); ++i) p.qpre[i] = f2bf((float)((double)bf2f(p.q[i]) * kScale * kLog2e));
    HC(hipMalloc(&p.dqpre, p.q.size() * 2));
    HC(hipMemcpy(p.dqpre, p.qpre.data(), p.q.size() * 2, hipMemcpyHostToDevice));
    HC(hipMalloc(&p.dq, p.q.size() * 2)); HC(hipMalloc(&p.dk, p.k.size() * 2)); HC(hipMalloc(&p.dvt, p.vt.size() * 2)); HC(hipMalloc(&p.d_o, p.o0.size() * 2));
    HC(hipMemcpy(p.dq, p.q.data(), p.q.size() * 2, hipMemcpyHostToDevice));
    HC(hipMemcpy(p.dk, p.k.data(), p.k.size() * 2, hipMemcpyHostToDevice));
    HC(hipMemcpy(p.dvt, p.vt.data(), p.vt.size() * 2, hipMemcpyHostToDevice));
    {
        const int64_t nk = (Lk + 63) / 64 * 64;
        p.ldvtp = nk;
        std::vector<uint16_t> kp(nk * p.ldk, 0x7fc0), vp(H * 128 * nk);
        memcpy(kp.data(), p.k.data(), p.k.size() * 2);
        for (int64_t r = 0; r < H * 128; ++r)
            for (int64_t c = 0; c < nk; ++c) vp[r * nk + c] = c < Lk ? p.vt[r * p.ldvt + c] : f2bf(100.0f * rnd());
        HC(hipMalloc(&p.dkp, kp.size() * 2)); HC(hipMalloc(&p.dvtp, vp.size() * 2));
        HC(hipMemcpy(p.dkp, kp.data(), kp.size() * 2, hipMemcpyHostToDevice));
        HC(hipMemcpy(p.dvtp, vp.data(), vp.size() * 2, hipMemcpyHostToDevice));
    }
    p.wsb = yume_attn_workspace_bytes(Lq, Lk, H);
    p.ws = nullptr;
    if (p.wsb) HC(hipMalloc(&p.ws, p.wsb));
}
static void drop(Prob& p) { hipFree(p.dqpre); hipFree(p.dq); hipFree(p.dk); hipFree(p.dvt); hipFree(p.d_o); hipFree(p.dkp); hipFree(p.dvtp); if (p.ws) hipFree(p.ws); }

static int run(Prob& p, int variant, int accumulate, std::vector<uint16_t>& out, bool scratch = true) {
    HC(hipMemcpy(p.d_o, p.o0.data(), p.o0.size() * 2, hipMemcpyHostToDevice));
    const bool pad = (variant & 512) != 0;
    int rc = yume_attn_fwd_ws((variant & 256) ? p.dqpre : p.dq, p.ldq, pad ? p.dkp : p.dk, p.ldk, pad ? p.dvtp : p.dvt, pad ? p.ldvtp : p.ldvt, p.d_o, p.ldo, p.Lq, p.Lk,
                              p.H, 0.08838834764831845f, accumulate, variant, scratch ? p.ws : nullptr, scratch ? p.wsb : 0, nullptr);
    if (rc) { printf("  variant %d: rc=%d %s\n", variant, rc, yume_last_error()); return rc; }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("  variant %d: device error %s\n", variant, hipGetErrorString(e)); exit(3); }
    out.resize(p.o0.size());
    HC(hipMemcpy(out.data(), p.d_o, out.size() * 2, hipMemcpyDeviceToHost));
    return 0;
}

static void reference(const Prob& p, int accumulate, std::vector<float>& ref, bool pre = false) {
    const std::vector<uint16_t>& Q = pre ? p.qpre : p.q;
    const double sc = pre ? 0.6931471805599453 : kScale;      // exp2(q' . k) = exp(ln 2 * q' . k)
    ref.assign(p.Lq * p.ldo, 0.f);
    std::vector<double> s(p.Lk);
    for (int64_t h = 0; h < p.H; ++h)
        for (int64_t i = 0; i < p.Lq; ++i) {
            double mx = -1e300;
            for (int64_t j = 0; j < p.Lk; ++j) {
                double a = 0;
                for (int d = 0; d < 128; ++d) a += (double)bf2f(Q[i * p.ldq + h * 128 + d]) * bf2f(p.k[j * p.ldk + h * 128 + d]);
                s[j] = a * sc;
                mx = s[j] > mx ? s[j] : mx;
            }
            double l = 0;
            for (int64_t j = 0; j < p.Lk; ++j) { s[j] = exp(s[j] - mx); l += s[j]; }
            for (int d = 0; d < 128; ++d) {
                double a = 0;
                for (int64_t j = 0; j < p.Lk; ++j) a += s[j] * bf2f(p.vt[(h * 128 + d) * p.ldvt + j]);
                a /= l;
                if (accumulate) a += bf2f(p.o0[i * p.ldo + h * 128 + d]);
                ref[i * p.ldo + h * 128 + d] = (float)a;
            }
        }
}

static double maxdiff(const std::vector<uint16_t>& a, const std::vector<float>& r, int* nan) {
    double m = 0; *nan = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        float x = bf2f(a[i]);
        if (x != x) { ++*nan; continue; }
        double d = fabs((double)x - r[i]);
        m = d > m ? d : m;
    }
    return m;
}
// where two outputs differ: how many values, in how many (query block, head) cells, the first few places — enough to tell a wrong item
// boundary from a wrong tile or a wrong mask
static size_t where_differs(const Prob& p, const std::vector<uint16_t>& a, const std::vector<uint16_t>& b) {
    size_t nd = 0, cells = 0;
    std::vector<char> cell(((p.Lq + 255) / 256) * p.H, 0);
    for (int64_t i = 0; i < p.Lq; ++i)
        for (int64_t c = 0; c < p.H * 128; ++c) {
            const size_t ix = i * p.ldo + c;
            if (a[ix] == b[ix]) continue;
            if (nd < 6) printf("      row %lld (block %lld, wave %lld, lane-row %lld) head %lld d %lld: %g vs %g\n", (long long)i, (long long)(i / 256), (long long)((i % 256) / 64),
                               (long long)(i % 64), (long long)(c / 128), (long long)(c % 128), bf2f(a[ix]), bf2f(b[ix]));
            ++nd;
            char& f = cell[(i / 256) * p.H + c / 128];
            if (!f) { f = 1; ++cells; }
        }
    if (nd) {
        printf("      %zu values differ in %zu of %zu (block, head) cells\n", nd, cells, cell.size());
        // by wave of the 256-query workgroup and by its two 32-query blocks (A = lane rows 0..31, B = 32..63), and by size in bf16 ulps of the larger value
        size_t byw[4][2] = {}, ulp[4] = {};
        for (int64_t i = 0; i < p.Lq; ++i)
            for (int64_t c = 0; c < p.H * 128; ++c) {
                const size_t ix = i * p.ldo + c;
                if (a[ix] == b[ix]) continue;
                ++byw[(i % 256) / 64][(i % 64) / 32];
                const float x = bf2f(a[ix]), y = bf2f(b[ix]), m = fmaxf(fabsf(x), fabsf(y));
                const float u = m > 0 ? fabsf(x - y) / (m * 0.0078125f) : 0.f;
                ++ulp[u <= 1.01f ? 0 : u <= 2.01f ? 1 : u <= 8.f ? 2 : 3];
            }
        printf("      by wave x block (A B): w0 %zu %zu | w1 %zu %zu | w2 %zu %zu | w3 %zu %zu;  <=1 ulp %zu, 2 ulp %zu, <=8 ulp %zu, more %zu\n", byw[0][0], byw[0][1],
               byw[1][0], byw[1][1], byw[2][0], byw[2][1], byw[3][0], byw[3][1], ulp[0], ulp[1], ulp[2], ulp[3]);
    }
    return nd;
}
static double maxdiff2(const std::vector<uint16_t>& a, const std::vector<uint16_t>& b, int* nan) {
    double m = 0; *nan = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        float x = bf2f(a[i]), y = bf2f(b[i]);
        if (x != x || y != y) { ++*nan; continue; }
        double d = fabs((double)x - y);
        m = d > m ? d : m;
    }
    return m;
}

int main(int argc, char** argv) {
    std::vector<int> variants;
    const char* lib = "yume_amd/lib/libyume_hip.so";
    bool timing_only = false, small_only = false;
    int big_spike = 1;
    int one[3] = {0, 0, 0};
    const char* trace = nullptr;      // --trace file: dump the per-workgroup time stamps of an experiment build (csrc/trace.hpp) after a --one run
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--lib")) lib = argv[++i];
        else if (!strcmp(argv[i], "--timing")) timing_only = true;
        else if (!strcmp(argv[i], "--nospike")) big_spike = 0;
        else if (!strcmp(argv[i], "--small")) small_only = true;
        else if (!strcmp(argv[i], "--trace")) trace = argv[++i];
        else if (!strcmp(argv[i], "--one")) { one[0] = atoi(argv[i + 1]); one[1] = atoi(argv[i + 2]); one[2] = atoi(argv[i + 3]); i += 3; timing_only = true; }
        else variants.push_back(atoi(argv[i]));
    }
    void* hnd = dlopen(lib, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
    if (!hnd) { printf("dlopen %s: %s\n", lib, dlerror()); return 2; }
    p_attn = (attn_fn)dlsym(hnd, "yume_attn_fwd_ws"); p_ws = (ws_fn)dlsym(hnd, "yume_attn_workspace_bytes"); p_err = (err_fn)dlsym(hnd, "yume_last_error");
    printf("library %s\n", lib);
    {   // the caller-owned ticket-counter workspace (include/yume_hip.h): the persistent attention kernel draws its items from it
        cwb_fn cwb = (cwb_fn)dlsym(hnd, "yume_counter_workspace_bytes");
        cwi_fn cwi = (cwi_fn)dlsym(hnd, "yume_counter_workspace_init");
        if (cwb && cwi) {
            void* cw = nullptr;
            HC(hipMalloc(&cw, cwb()));
            if (cwi(cw, cwb(), nullptr)) { printf("counter workspace: %s\n", yume_last_error()); return 2; }
            HC(hipDeviceSynchronize());
        }
    }
    if (variants.empty()) variants = {7, 263, 256, 776, 768, 4, 2, 0};
    int fails = 0;
    const int small[][4] = {{256, 64, 1, 0}, {64, 40, 1, 0}, {1, 1, 1, 0}, {300, 200, 2, 0}, {273, 323, 3, 0}, {256, 256, 1, 0}, {256, 320, 1, 0},
                            {513, 640, 9, 0}, {700, 1000, 8, 1}, {256, 577, 2, 1}, {260, 448, 1, 0}, {512, 512, 3, 1}, {384, 1999, 2, 1},
                            {700, 2100, 8, 1}, {1000, 3333, 16, 1}, {300, 1536, 1, 0}, {300, 1600, 2, 2}, {700, 2100, 8, 2}, {512, 4096, 3, 2},
                            {256, 64, 1, 2}, {1100, 1984, 8, 0}, {2000, 512, 3, 0}, {1300, 515, 9, 1}, {4000, 2500, 12, 2}, {9000, 300, 2, 0}};
    for (auto& sh : small) {
        if (timing_only) break;
        for (int acc = 0; acc < 2; ++acc) {
            Prob p; make(p, sh[0], sh[1], sh[2], sh[3]);
            std::vector<float> ref, refpre; reference(p, acc, ref);
            bool anypre = false;
            for (int v : variants) anypre |= (v & 256) != 0;
            if (anypre) reference(p, acc, refpre, true);
            for (int v : variants) {
                if ((v == 4) && (sh[0] < 1)) continue;
                if ((v & 255) == 8 && (sh[0] < 256 || sh[1] < 512)) continue;          // outside the persistent kernel's shapes (it says so: checked below)
                std::vector<uint16_t> out;
                if (run(p, v, acc, out)) { ++fails; continue; }
                if ((v & 255) == 8) {       // without scratch (whole query blocks only) the persistent kernel and variant 7 must agree bit for bit
                    std::vector<uint16_t> a8, a7, a8b;
                    if (run(p, v, acc, a8, false) || run(p, 7 | 256, acc, a7, false) || run(p, v, acc, a8b, false)) { ++fails; continue; }
                    if (a8 != a8b) { printf("      variant %d is NOT deterministic:\n", v); where_differs(p, a8, a8b); }
                    const size_t nd = where_differs(p, a8, a7);
                    printf("small Lq=%d Lk=%d H=%d spike=%d acc=%d variant=%d  vs variant 263 without scratch: %zu differing values %s\n", sh[0], sh[1], sh[2], sh[3], acc, v, nd,
                           nd ? "FAIL" : "ok");
                    if (nd) ++fails;
                }
                int nan; double md = maxdiff(out, (v & 256) ? refpre : ref, &nan);
                const bool ok = nan == 0 && md < (acc ? 6e-2 : 4e-2);
                printf("small Lq=%d Lk=%d H=%d spike=%d acc=%d variant=%d  maxabs=%.3e nan=%d %s\n", sh[0], sh[1], sh[2], sh[3], acc, v, md, nan, ok ? "ok" : "FAIL");
                if (!ok) ++fails;
            }
            drop(p);
        }
    }
    const int big[][3] = {{9460, 9460, 24}, {8192, 9460, 24}, {9460, 512, 24}, {2048, 4096, 16}, {23460, 23460, 40}, {27810, 27810, 40}};
    if (one[0]) {      // a single problem, the listed variants only (profiling runs)
        Prob p; make(p, one[0], one[1], one[2], 0);
        for (int v : variants) {
            std::vector<uint16_t> out;
            for (int i = 0; i < 3; ++i) run(p, v, 0, out);
        }
        if (trace) {
            typedef int (*trace_fn)(void*, long long);
            trace_fn rd = (trace_fn)dlsym(hnd, "yume_debug_trace_read");
            if (!rd) { printf("--trace: %s is not a -DYUME_TRACE build\n", lib); return 2; }
            std::vector<unsigned long long> t(32768 * 8);
            HC(hipDeviceSynchronize());
            if (rd(t.data(), (long long)t.size() * 8)) { printf("trace read failed\n"); return 2; }
            FILE* f = fopen(trace, "wb"); fwrite(t.data(), 8, t.size(), f); fclose(f);
            printf("trace written to %s\n", trace);
        }
        drop(p);
        return 0;
    }
    int nbig = 0;
    for (auto& sh : big) {
        if (small_only) break;
        if (timing_only && nbig++ >= 3) break;
        Prob p; make(p, sh[0], sh[1], sh[2], big_spike);
        std::vector<uint16_t> base, basepre;
        if (run(p, 2, 0, base)) { ++fails; drop(p); continue; }
        if (run(p, 2 | 256, 0, basepre)) { ++fails; drop(p); continue; }
        for (int v : variants) {
            std::vector<uint16_t> out;
            if (run(p, v, 0, out)) { ++fails; continue; }
            int nan; double md = maxdiff2(out, (v & 256) ? basepre : base, &nan);
            hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
            const int it = sh[0] > 20000 ? 3 : 10;
            const uint16_t* qd = (v & 256) ? p.dqpre : p.dq;
            const uint16_t *kd = (v & 512) ? p.dkp : p.dk, *vd = (v & 512) ? p.dvtp : p.dvt;
            const int64_t ldv = (v & 512) ? p.ldvtp : p.ldvt;
            for (int i = 0; i < 2; ++i) yume_attn_fwd_ws(qd, p.ldq, kd, p.ldk, vd, ldv, p.d_o, p.ldo, p.Lq, p.Lk, p.H, 0.0883883f, 0, v, p.ws, p.wsb, nullptr);
            HC(hipEventRecord(e0, nullptr));
            for (int i = 0; i < it; ++i) yume_attn_fwd_ws(qd, p.ldq, kd, p.ldk, vd, ldv, p.d_o, p.ldo, p.Lq, p.Lk, p.H, 0.0883883f, 0, v, p.ws, p.wsb, nullptr);
            HC(hipEventRecord(e1, nullptr)); HC(hipEventSynchronize(e1));
            float ms; HC(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
            const double tf = 4.0 * sh[0] * sh[1] * 128.0 * sh[2] / (ms * 1e-3) / 1e12;
            const bool ok = timing_only || (nan == 0 && md < 3e-2);
            printf("big Lq=%d Lk=%d H=%d variant=%d  vs v2 maxabs=%.3e nan=%d %s   %.3f ms  %.0f TFLOP/s\n", sh[0], sh[1], sh[2], v, md, nan, ok ? "ok" : "FAIL", ms, tf);
            if (!ok) ++fails;
            if ((v & 255) == 8 && !timing_only) {
                std::vector<uint16_t> a8, a7;
                if (run(p, v, 0, a8, false) || run(p, 7 | 256, 0, a7, false)) { ++fails; continue; }
                const size_t nd = where_differs(p, a8, a7);
                printf("big Lq=%d Lk=%d H=%d variant=%d  vs variant 263 without scratch: %zu differing values %s\n", sh[0], sh[1], sh[2], v, nd, nd ? "FAIL" : "ok");
                if (nd) ++fails;
            }
        }
        drop(p);
    }
    printf("attn_check: %d failure(s)\n", fails);
    return fails ? 1 : 0;
}
