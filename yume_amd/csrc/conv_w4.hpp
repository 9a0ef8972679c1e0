// conv_w4.hpp — the implicit-GEMM convolution on the one-wave-per-SIMD pipeline of gemm_w4.hpp (r3).
//
// The 8-wave kernel's conv loader (conv3d.hip, ConvAFast) computes a 64-bit source pointer per row and per K tile — mask test, two selects,
// a 64-bit add, for each of its 4 rows, in VALU instructions beside the MFMAs: its matrix pipe is 49 % busy on the dominant Wan2.2-decoder
// convolution (r2 PMC) against 72 % of the same kernel on a dense GEMM. Here, for stride-1 convolutions whose frames are whole tiles
// (Ho*Wo % 256 == 0; Cin % 64 == 0, no folded upsample):
//   * a row's source offset is loop-invariant: va[j] = (ho*Win + wo) * ldc*2 + swizzled chunk, relative to the input FRAME the tap reads;
//   * the tap is wave-uniform: the frame (x or the 2-frame causal cache) is a buffer descriptor in SGPRs, rebuilt when dt changes; the
//     position inside the frame is the SGPR offset soff = (dh*Win + dw) * ldc*2 + channel tile (descriptor base = frame - (ph*Win + pw)*ldc*2);
//   * a tap outside the image for a row (bit dh*kw+dw of the row's mask clear) is one v_cndmask: the lane's offset becomes 0xffffffff,
//     beyond the descriptor's range, and the LDS-DMA writes zeros for it — no zero page, no pointer select; a missing cache is a
//     descriptor of zero records;
//   * K is walked (dt, channel tile, dh, dw) as in conv3d.hip (L2 reuse across the kh*kw taps); the weight tile of a position is W + wk.
// Everything else — LDS images, fragment reads, gap plan, epilogues — is gemm_w4.hpp's.
#pragma once
#include "gemm_w4.hpp"
#include "counters.hpp"

namespace gemm_w4 {

struct ConvW4 {
    const unsigned short* x;       // [Tin, Hin, Win, ldc]
    const unsigned short* cache;   // [2, Hin, Win, ldc] or nullptr
    int64_t ldc;
    int Tin, Hin, Win, Cin, To, Ho, Wo;
    int kt, kh, kw, pt, ph, pw;
    int ups;                       // the conv reads the nearest-2x upsampled view of x (Ho = 2 Hin, Wo = 2 Win)
    // Long launches (>= 8 rounds of tiles): the XCDs of one chip do not run the same code equally fast (the starts of a round drift apart
    // by about a tile time per 12 rounds, profiles/r3_gemm_w4.md section 7), and a static split leaves the fast ones idle at the end. The
    // last `dyn` tiles of every XCD's chunk are handed out by ticket (steal[xcd]): first to the XCD's own workgroups, then to workgroups
    // of XCDs whose own tickets have run out — among them 32 surplus workgroups per XCD that exist only to steal. steal = one set of the
    // caller's counter workspace (counters.hpp: zeros between launches; steal[8] counts the workgroups that have drawn, the last of the
    // `nticket` puts the zeros back). nullptr: the static split.
    int* steal;
    int dyn;
    int nticket;
};

// wave-uniform walk of the K tiles (the tile to STAGE): order (dt, channel tile, dh, dw)
struct ConvWalk {
    int dt, dh, dw, cin;
    int wk;               // element offset of the K tile inside a weight row
    unsigned soff;        // byte offset of tap (dh, dw) + channel tile inside the frame
};

__device__ __forceinline__ void conv_set_frame(Ctx& c, const ConvW4& cv, int to, int dt) {
    const int ti = to - cv.pt + dt;                                  // (stride 1)
    const int64_t frame = (int64_t)cv.Hin * cv.Win * cv.ldc;         // elements
    const unsigned short* base = ti >= 0 ? cv.x + (int64_t)ti * frame : cv.cache + (int64_t)(ti + 2) * frame;
    const bool have = (ti >= 0 && ti < cv.Tin) || (ti < 0 && ti >= -2 && cv.cache != nullptr);
    // the descriptor starts (ph rows + pw positions) in front of the frame; the rows' offsets va[] are biased by the same amount
    const uint64_t b = (uint64_t)(uintptr_t)base - (uint64_t)((int64_t)(cv.ph * cv.Win + cv.pw) * cv.ldc * 2);
    i32x4 d;
    d[0] = (int)(unsigned)(b & 0xffffffffu);
    d[1] = (int)(unsigned)((b >> 32) & 0xffffu);                    // stride 0: raw buffer
    d[2] = have ? 0x7fffffff : 0;                                    // num_records: every in-image offset passes, 0xffffffff never does
    d[3] = 0x00020000;
    c.srd = d;
}

template <int CONV>
__device__ __forceinline__ void conv_apply(Ctx& c, const ConvW4& cv, const ConvWalk& w, const char* pbw) {
    c.tapmask = 1u << (w.dh * cv.kw + w.dw);
    c.pb = pbw + (int64_t)w.wk * 2;
    if constexpr (CONV == 2) {
        // upsampled row 2a+e under tap dh (ph = 1) reads input row a + floor((e + dh - 1) / 2): e = 0: (-1, 0, 0), e = 1: (0, 0, +1); same for columns
        const unsigned ldc2 = (unsigned)cv.ldc * 2u, rowb = (unsigned)cv.Win * ldc2;
        c.soff = (unsigned)w.cin * 2u;
        c.uh[0] = w.dh == 0 ? 0u - rowb : 0u;
        c.uh[1] = w.dh == 2 ? rowb : 0u;
        c.uw[0] = w.dw == 0 ? 0u - ldc2 : 0u;
        c.uw[1] = w.dw == 2 ? ldc2 : 0u;
    } else {
        c.soff = w.soff;
    }
}

// one step of the walk; returns true when dt changed (new frame descriptor)
__device__ __forceinline__ bool conv_advance(ConvWalk& w, const ConvW4& cv) {
    const unsigned ldc2 = (unsigned)cv.ldc * 2u;
    if (++w.dw < cv.kw) {
        w.wk += cv.Cin;
        w.soff += ldc2;
        return false;
    }
    w.dw = 0;
    w.wk -= (cv.kw - 1) * cv.Cin;
    w.soff -= (unsigned)(cv.kw - 1) * ldc2;
    if (++w.dh < cv.kh) {
        w.wk += cv.kw * cv.Cin;
        w.soff += (unsigned)cv.Win * ldc2;
        return false;
    }
    w.dh = 0;
    w.wk -= (cv.kh - 1) * cv.kw * cv.Cin;
    w.soff -= (unsigned)(cv.kh - 1) * (unsigned)cv.Win * ldc2;
    w.cin += BK;
    w.wk += BK;
    w.soff += 2 * BK;
    if (w.cin < cv.Cin) return false;
    w.cin = 0;
    w.soff -= 2u * (unsigned)cv.Cin;
    w.wk += (cv.kh * cv.kw - 1) * cv.Cin;
    ++w.dt;
    return true;
}

template <bool DMA, int CONV>
__device__ __forceinline__ void conv_w4_loop(Ctx& c, const ConvW4& cv, ConvWalk& w, const char* pbw, int to, int t0, int t1, unsigned lbase) {
    for (int t = t0; t < t1; ++t) {
        w4_gaps<0, DMA, true, CONV>(c);
        if constexpr (DMA) {
            if (conv_advance(w, cv)) conv_set_frame(c, cv, to, w.dt);
            conv_apply<CONV>(c, cv, w, pbw);
        }
        c.lcur = lbase + (((t + 1) & 1) ? BUF_BYTES : 0);
        c.ra1 ^= BUF_BYTES;
        c.rb1 ^= BUF_BYTES;
        c.ra0n ^= BUF_BYTES;
        c.rb0n ^= BUF_BYTES;
    }
}

// to: output frame of the tile, r0: its first position inside the frame (a tile never spans two frames: a frame of Ho*Wo positions takes
// ceil(Ho*Wo / 256) tiles, the rows of its last tile beyond the frame read nothing — every tap masked — and are not stored)
template <int CONV>
__device__ __forceinline__ void conv_w4_mainloop(const Problem& p, const ConvW4& cv, char* smem, int to, int r0, int n0) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    Ctx c;
    c.smem = smem;
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;
    c.lcur = lbase;
    c.pa = nullptr;
    const char* pbw = reinterpret_cast<const char*>(p.W + (int64_t)n0 * p.ldw);
    const unsigned ldc2 = (unsigned)cv.ldc * 2u, ldw_b = (unsigned)p.ldw * 2u;
    const int HW = cv.Ho * cv.Wo;
    {
        const int rl = 8 * wave + (lane >> 3);
        const unsigned ch = (unsigned)(((lane & 7) ^ (rl & 7)) << 4);
        const int nleft = p.N - 1 - n0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 32 * j + rl;
            const int pos = r0 + r;
            const int ho = pos / cv.Wo, wo = pos - ho * cv.Wo;
            const int hh = ho - cv.ph, ww = wo - cv.pw;
            unsigned mk = 0;
            if constexpr (CONV == 2) {
                // offsets relative to the descriptor base = frame - (ph*Win + pw) positions (ph = pw = 1)
                c.va[j] = (unsigned)(((ho >> 1) + cv.ph) * cv.Win + (wo >> 1) + cv.pw) * ldc2 + ch;
                c.par[j] = (unsigned)((ho & 1) | ((wo & 1) << 1));
                for (int a = 0; a < cv.kh; ++a)
                    for (int b = 0; b < cv.kw; ++b)
                        if (hh + a >= 0 && hh + a < 2 * cv.Hin && ww + b >= 0 && ww + b < 2 * cv.Win) mk |= 1u << (a * cv.kw + b);
            } else {
                c.va[j] = (unsigned)(ho * cv.Win + wo) * ldc2 + ch;
                for (int a = 0; a < cv.kh; ++a)
                    for (int b = 0; b < cv.kw; ++b)
                        if (hh + a >= 0 && hh + a < cv.Hin && ww + b >= 0 && ww + b < cv.Win) mk |= 1u << (a * cv.kw + b);
            }
            c.mask[j] = pos < HW ? mk : 0u;
            c.vb[j] = (unsigned)min(r, nleft) * ldw_b + ch;
        }
    }
    const unsigned sw0 = (unsigned)(((lane >> 4) ^ (lane & 7)) << 4), sw1 = (unsigned)(((4 | (lane >> 4)) ^ (lane & 7)) << 4);
    const unsigned rowa = wr * 16384 + (lane & 15) * 128, rowb = OPER_BYTES + wc * 16384 + (lane & 15) * 128;
    const int nk = p.K / BK;
    ConvWalk w = {0, 0, 0, 0, 0, 0u};
    conv_set_frame(c, cv, to, 0);
    conv_apply<CONV>(c, cv, w, pbw);
    // ---- prologue: tiles 0 and 1 in flight, accumulators zeroed under their latency, F0(0) read ----
    w4_stage_all<0, CONV>(c);
    if (conv_advance(w, cv)) conv_set_frame(c, cv, to, w.dt);
    conv_apply<CONV>(c, cv, w, pbw);
    c.lcur = lbase + BUF_BYTES;
    w4_stage_all<0, CONV>(c);
    if (conv_advance(w, cv)) conv_set_frame(c, cv, to, w.dt);
    conv_apply<CONV>(c, cv, w, pbw);
    c.lcur = lbase;
    acc_zero<0, 256>();
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        c.fb[0][r] = w4_frag(c, rowb + sw0 + r * 2048);
        c.fa[0][r] = w4_frag(c, rowa + sw0 + r * 2048);
    }
    c.ra1 = rowa + sw1;
    c.rb1 = rowb + sw1;
    c.ra0n = BUF_BYTES + rowa + sw0;
    c.rb0n = BUF_BYTES + rowb + sw0;
    conv_w4_loop<true, CONV>(c, cv, w, pbw, to, 0, nk - 2, lbase);
    conv_w4_loop<false, CONV>(c, cv, w, pbw, to, nk - 2, nk, lbase);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
}

template <int UNUSED = 0>
__global__ __launch_bounds__(NTHR_W4, 1) void conv_w4_kernel(Problem p, ConvW4 cv, Epilogue e, int epi) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_W4];
    int start, count, m0, n0;
    TRACE_STAMP(0);
    const int ntile = p.tiles_m * p.tiles_n, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    xcd_chunk(ntile, xcd, start, count);
    int tile = start + idx;
    if (cv.steal != nullptr && idx >= count - cv.dyn) {            // (workgroup-uniform) one of the ticketed tiles, or a surplus workgroup
        int* slot = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) {
            int got = -1;
            const int dx = min(count, cv.dyn), t = atomicAdd(&cv.steal[xcd], 1);
            if (t < dx) got = start + count - dx + t;
            for (int k = 1; k < 8 && got < 0; ++k) {
                const int y = (xcd + k) & 7;
                int sy, cy;
                xcd_chunk(ntile, y, sy, cy);
                const int dy = min(cy, cv.dyn), ty = atomicAdd(&cv.steal[y], 1);
                if (ty < dy) got = sy + cy - dy + ty;
            }
            // The counter set is caller-owned memory that holds zeros between launches (counters.hpp): the LAST workgroup to have drawn
            // its tickets — nobody touches the set after it — puts the zeros back. No memset in front of the launch, nothing shared
            // with the next launch that is handed this set.
            __threadfence();
            if (atomicAdd(&cv.steal[8], 1) == cv.nticket - 1) {
#pragma unroll
                for (int k = 0; k < 9; ++k) atomicExch(&cv.steal[k], 0);
            }
            *slot = got;
        }
        __syncthreads();
        tile = *slot;
        __syncthreads();                                           // (the K loop's first LDS-DMA piece lands on the slot)
        if (tile < 0) return;
    }
    tile_origin(p, tile, m0, n0);                 // m0 = 256 x (row of tiles); p.tiles_m = To x tiles per frame
    const int HW = cv.Ho * cv.Wo, tpf = (HW + 255) >> 8, tm = m0 >> 8;
    const int to = tm / tpf, r0 = (tm - to * tpf) << 8;
    if (cv.ups) conv_w4_mainloop<2>(p, cv, smem, to, r0, n0);
    else conv_w4_mainloop<1>(p, cv, smem, to, r0, n0);
    TRACE_STAMP(1);
    // the epilogue sees the tile's frame as the end of the matrix: output rows [to*HW + r0, (to+1)*HW) — whole tiles when HW % 256 == 0
    Problem pe = p;
    pe.M = (to + 1) * HW;
    m0 = to * HW + r0;
    switch (epi) {
        case YUME_EPI_F32: w4_epilogue<YUME_EPI_F32, true>(pe, e, m0, n0, smem); break;
        case EPI_BF16_ADD: w4_epilogue<EPI_BF16_ADD, true>(pe, e, m0, n0, smem); break;
        case EPI_BF16_TSPLIT: w4_epilogue<EPI_BF16_TSPLIT, true>(pe, e, m0, n0, smem); break;
        default: w4_epilogue<YUME_EPI_BF16, true>(pe, e, m0, n0, smem); break;
    }
    TRACE_STAMP(2);
}

// what the kernel takes (host): stride 1, Cin in whole K tiles, frames of at least 1024 positions (a frame takes ceil(Ho*Wo / 256) tiles:
// at most 20 % of the rows of its tiles idle; the decoder's 44 x 80 level: 1.8 %), >= 3 K tiles, 32-bit offsets
inline bool conv_w4_applies(const Problem& p, const ConvW4& cv, int st, int sh, int sw, int ups, int epi) {
    static const bool on = [] { const char* v = getenv("YUME_CONV_W4"); return !v || atoi(v) != 0; }();
    if (!on || st != 1 || sh != 1 || sw != 1) return false;
    if (ups && (cv.kt != 1 || cv.kh != 3 || cv.kw != 3 || cv.ph != 1 || cv.pw != 1 || cv.pt != 0 || cv.Ho != 2 * cv.Hin || cv.Wo != 2 * cv.Win)) return false;
    // (the folded-upsample source keeps whole-tile frames: at few frames per pass its launches fall to the gathering kernel, which sums the
    // taps in the weight's own order — a ragged level on this kernel only when many frames are batched would make the one-latent-per-pass
    // walk and the grouped pass of the decoder differ in their last bits, tests/test_live_fullsize_gpu.py)
    if (ups && ((int64_t)cv.Ho * cv.Wo) % 256 != 0) return false;
    if ((cv.Cin % BK) != 0 || (int64_t)cv.Ho * cv.Wo < 1024 || (int64_t)cv.Ho * cv.Wo * cv.To >= (1ll << 31) || cv.kh * cv.kw > 32 || p.K < 3 * BK) return false;
    if ((int64_t)cv.Hin * cv.Win * cv.ldc * 2 + (int64_t)(cv.kh * cv.Win + cv.kw) * cv.ldc * 2 >= 0x7fffff00ll) return false;
    if (255ll * p.ldw * 2 + 128 >= (1ll << 32)) return false;
    if (cv.pt > 2 || (int64_t)p.tiles_m * p.tiles_n * 4 < 192) return false;          // too few tiles to fill the chip: the 128x128 kernel's job
    if (epi == EPI_BF16_TSPLIT && (((p.N >> 1) % 256) != 0)) return false;
    return epi == YUME_EPI_BF16 || epi == YUME_EPI_F32 || epi == EPI_BF16_ADD || epi == EPI_BF16_TSPLIT;
}

inline int launch_conv_w4(int epi, const Problem& p128, const ConvW4& cv, const Epilogue& e, hipStream_t st, const char* what) {
    Problem p = p128;
    p.tiles_m = cv.To * ((cv.Ho * cv.Wo + 255) / 256);          // whole frames of tiles
    p.tiles_n = (p.N + 255) / 256;
    p.group_m = g_group_m;
    p.epi_direct = 0;
    const int ntile = p.tiles_m * p.tiles_n;
    ConvW4 c = cv;
    c.steal = nullptr;
    c.dyn = 0;
    c.nticket = 0;
    unsigned grid = (unsigned)ntile;
    static const bool steal_on = [] { const char* v = getenv("YUME_CONV_STEAL"); return !v || atoi(v) != 0; }();
    if (steal_on && ntile >= 8 * 256) {
        // ticket counters: one 64-byte set of the caller's counter workspace (yume_counter_workspace_init; counters.hpp) — the library
        // allocates nothing. Without a registered workspace the launch keeps its static tile order.
        c.steal = yume_counters::next_set();
        if (c.steal) {
            c.dyn = 96;
            grid += 8 * 32;
            c.nticket = 0;                                 // workgroups that will draw a ticket: per XCD the last `dyn` tiles + the surplus ids
            for (int x = 0; x < 8; ++x) {
                int sx, cx;
                xcd_chunk(ntile, x, sx, cx);
                const int blocks_x = ((int)grid - x + 7) / 8, first = cx > c.dyn ? cx - c.dyn : 0;
                c.nticket += blocks_x - first;
            }
        }
    }
    hipLaunchKernelGGL(conv_w4_kernel<0>, dim3(grid), dim3(NTHR_W4), 0, st, p, c, e, epi);
    YUME_CHECK_LAUNCH(what);
    return YUME_OK;
}

}  // namespace gemm_w4
