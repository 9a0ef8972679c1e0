/*
 * yume_hip.h — C-ABI of libyume_hip.so: the MI355X (gfx950) kernels behind YUME's
 * denoise hot path (WanModel DiT block stack + causal 3D VAE).
 *
 * The reference has no FFI of its own for this path: it is Python calling torch ops and
 * one external native library (flash-attn).  Each entry point below therefore names the
 * reference Python call site(s) it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory owned by the caller (torch-allocated);
 *     the library never allocates or frees device memory. The only device memory it keeps a
 *     pointer to between calls is the ticket-counter workspace the caller registers with
 *     yume_counter_workspace_init (below); without one, every kernel runs a ticket-free schedule.
 *   - shapes are element counts, strides/leading dimensions are in ELEMENTS of the buffer type.
 *   - bf16 buffers are passed as `const void*` / `void*` (raw 16-bit brain floats).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream). No entry point
 *     synchronises; all work is enqueued on `stream`.
 *   - return value: 0 on success, a negative YUME_E* code on failure; the message for the
 *     last failure on the calling thread is returned by yume_last_error(). Nothing throws
 *     across the ABI.
 *   - re-entrant across processes (one process per GPU); no internal threads, no global
 *     mutable device state.
 */
#ifndef YUME_HIP_H
#define YUME_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YUME_OK 0
#define YUME_EINVAL (-1)   /* bad argument (shape/alignment/NULL)          */
#define YUME_ELAUNCH (-2)  /* hipLaunchKernel / runtime error               */
#define YUME_EUNSUP (-3)   /* combination not implemented by this build     */

/* ---- library info ------------------------------------------------------------------- */
const char* yume_last_error(void);
/* ABI version of this header; bumped on any signature change (yume_amd/_lib.py refuses a library that reports another one). */
#define YUME_ABI_VERSION 8
int yume_abi_version(void);
/* name of the gfx target the kernels were compiled for ("gfx950"). */
const char* yume_target_arch(void);

/* ---- caller-owned ticket-counter workspace --------------------------------------------------
 * Kernels that hand out work by ticket (the tails of long convolutions; the persistent attention kernel) draw their counters from a
 * buffer the CALLER owns: yume_counter_workspace_bytes() bytes, 64-byte aligned, registered once per device with
 * yume_counter_workspace_init(ptr, bytes, stream) — the call zeroes it on `stream` — and valid until it is replaced or unregistered
 * (ptr = NULL). Invariant: the buffer holds zeros whenever no launch is using it (the last workgroup of a launch to touch its 64-byte
 * counter set writes the zeros back), so launches need no memset and can be captured into a hipGraph (the set a launch uses is chosen when
 * the launch is enqueued — at capture time for a graph: two replays of ONE captured graph must not run concurrently, as for any graph whose
 * kernels share scratch). Otherwise a set is shared by two launches only if 256 ticketed launches of one device are in flight at once. Without a registered buffer the ticketed kernels keep their static
 * schedules (same results). The registration is per process and per device (the calling thread's current device). */
int64_t yume_counter_workspace_bytes(void);
int yume_counter_workspace_init(void* ptr, int64_t bytes, void* stream);

/* ---- per-box calibration (r5; no reference counterpart — measurement plumbing of bench.py) ----------------
 * Launches `workgroups` x 256 threads (one wave per SIMD), each wave issuing iters * 16 v_mfma_f32_32x32x16_bf16 (32768 flop each, 32
 * matrix-pipe clocks each) on random operands (constant ones draw little power and run at 2.3 GHz) and nothing else. The caller times the launch with events on `stream`: workgroups * 4 * iters * 16 * 32768
 * flop / time = the dense bf16 rate THIS chip sustains under a pure matrix load (nominal 2.5 PFLOP/s assumes 2.4 GHz; the package power
 * limit holds a loaded MI355X near 1.7-1.8 GHz, and by how much differs from box to box). ticks (optional): uint64 [workgroups][2] =
 * {s_memtime delta, s_memrealtime delta (100 MHz)} of each workgroup's first wave; sink: 4 bytes of device memory (never written). */
int yume_calibrate_mfma(int64_t iters, int64_t workgroups, void* ticks, void* sink, void* stream);

/* ---- fused LayerNorm + modulate  ---------------------------------------------------------
 * replaces: wan23/modules/model.py:300-301,309-310 (norm1/norm2 + `*(1+scale)+shift`),
 *           wan23/modules/model.py:140-150 (WanLayerNorm), :308 (norm3, affine),
 *           wan23/modules/model.py:344-347 (Head norm+modulate);  same lines in wan/modules/model.py.
 *
 *   y[t, :] = LN(x[t, :]; eps, no affine) * (mul[row(t), :] + add_one) + add[row(t), :]
 *   row(t) = row_idx ? row_idx[t] : 0
 * x: fp32 [T, C] (row stride ldx).  mul/add: fp32 tables, row stride `tab_stride` elements
 * (for the DiT `mul`=scale chunk, `add`=shift chunk of the [R,6,C] modulation table with
 * add_one=1; for the affine norm3 `mul`=weight, `add`=bias with add_one=0).
 * out_kind: 0 = bf16 [T, C] (ldo), 1 = fp32 [T, C] (ldo),
 *           2 = bf16 3-way split [T, 3C]: [hi | hi | lo] where hi=bf16(y), lo=bf16(y-hi)
 *               (feeds the fp32-accurate head GEMM, see yume_gemm_bf16).
 * C must be a multiple of 8 and <= 8192.
 */
int yume_adaln_modulate(const float* x, int64_t ldx, int64_t T, int64_t C, float eps,
                        const float* mul, const float* add, int64_t tab_stride,
                        const int32_t* row_idx, int add_one,
                        void* out, int64_t ldo, int out_kind, void* stream);

/* ---- bf16 MFMA GEMM with fused epilogues ------------------------------------------------
 * replaces: every nn.Linear on the path — wan23/modules/model.py:171-174,189-195,205-206
 *           (q,k,v,o), :222-231 (cross q,k,v,o), :265-267,309-312 (ffn + gate/residual),
 *           :303-304 (gate + residual), :455-457,815-821 (text_embedding),
 *           :453-454,602-720 (patch embeddings as GEMM over gathered patches), :331,346 (head).
 *
 *   acc[m, n] = sum_k A[m, k] * W[n, k]          (A bf16 [M,K] lda;  W bf16 [N,K] ldw = nn.Linear layout)
 * K must be a multiple of 64; M, N arbitrary (> 0).
 * epilogue `epi`:
 *   YUME_EPI_BF16      out bf16 [M,N] (ldo)      = acc + bias
 *   YUME_EPI_BF16_GELU out bf16 [M,N]            = gelu_tanh(acc + bias)
 *   YUME_EPI_BF16_GELU_ERF  same with the exact erf GELU (img_emb MLPProj, wan/modules/model.py:534-537)
 *   YUME_EPI_F32       out fp32 [M,N]            = acc + bias
 *   YUME_EPI_RESID     out fp32 [M,N] in/out     += (acc + bias) * (gate ? gate[row(m), n] : 1)
 *                      gate fp32 table, row stride gate_stride, row(m) = row_idx ? row_idx[m] : 0
 *   YUME_EPI_BF16_SPLITT  columns n < n_split as YUME_EPI_BF16 into `out`;
 *                      columns n >= n_split written TRANSPOSED as bf16 into outT[(n - n_split), m]
 *                      (row stride ldt >= M) — the K-major V^T image the attention kernel consumes.
 *                      n_split must be a multiple of 128.
 * bias: fp32 [N] or NULL.
 * variant: 0 = automatic — a 256x256-tile kernel once its tiles fill the chip (r3: the one-wave-per-SIMD kernel of csrc/gemm_w4.hpp
 *   where it applies: K >= 192, the six epilogues above; env YUME_GEMM_W4=0 keeps the 8-wave kernel), else the 128x128-tile kernel;
 *   when the 256x256 tiling would leave a last round of tiles at most ~1/3 full, whole rounds of M-tiles go to the 256x256 kernel
 *   and the remaining rows to the 128x128 kernel (two launches inside this call). 1 = 128x128 kernel, 2 = the 8-wave 256x256
 *   kernel, 3 = the one-wave-per-SIMD 256x256 kernel (as 2 where it does not apply). All variants compute the same products in
 *   fp32; they differ in summation order only.
 */
enum {
    YUME_EPI_BF16 = 0,
    YUME_EPI_BF16_GELU = 1,
    YUME_EPI_F32 = 2,
    YUME_EPI_RESID = 3,
    YUME_EPI_BF16_SPLITT = 4,
    YUME_EPI_BF16_GELU_ERF = 5,
    YUME_EPI_BF16_GEGLU = 6      /* see the T5 section below */
};
int yume_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                   int64_t M, int64_t N, int64_t K, int epi,
                   void* out, int64_t ldo,
                   const float* gate, int64_t gate_stride, const int32_t* row_idx,
                   void* outT, int64_t ldt, int64_t n_split,
                   int variant, void* stream);

/* Same call with caller-owned scratch for the STREAM-K TAIL of the one-wave-per-SIMD kernel (r6, ABI 8): a GEMM whose 256 x 256 tiles are
 * not whole rounds of the chip's CUs (the block's N = 3072 projections: 444 tiles = 1.73 rounds) walks the whole rounds one tile per
 * workgroup and cuts the tiles of the last round(s) along K over one more full round of workgroups; partial tiles meet in `workspace`
 * (fp32, summed in a fixed order: results are run-to-run identical). workspace: yume_gemm_workspace_bytes() bytes, 16-byte aligned, ZERO
 * when first handed over (a launch returns every flag word to zero: no memset between launches, capturable into a hipGraph); launches
 * that share one workspace must be ordered on one stream (yume_amd/ops.py keeps one per device and stream). workspace = NULL or too
 * small: the schedules without it (whole tiles; row split + remainder launch), i.e. exactly yume_gemm_bf16. The word behind the last
 * flag (byte offset yume_gemm_workspace_bytes() - 64) is an error word: non-zero after a launch whose finisher timed out on a flag. */
int64_t yume_gemm_workspace_bytes(void);
int yume_gemm_bf16_ws(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                      int64_t M, int64_t N, int64_t K, int epi,
                      void* out, int64_t ldo,
                      const float* gate, int64_t gate_stride, const int32_t* row_idx,
                      void* outT, int64_t ldt, int64_t n_split,
                      int variant, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- RMSNorm over the hidden dim (+ optional 3D RoPE), in place, bf16 -----------------------
 * replaces: wan23/modules/model.py:121-137 (WanRMSNorm on q,k over the FULL hidden dim C),
 *           :38-118 (rope_apply: interleaved-pair complex multiply with the per-token table).
 *
 * buf: bf16 [T, nparts*C] (row stride ld); part p (p < nparts) = columns [p*C, (p+1)*C):
 *   y = x * rsqrt(mean_C(x^2) + eps) * w_p[c]        (fp32 math)
 *   if rope: head-wise (head_dim D, pairs (2j,2j+1)):  (y0,y1) <- (y0*cos - y1*sin, y0*sin + y1*cos)
 *            with rope = fp32 [T, D/2, 2] (cos, sin) per token, shared by all heads.
 * w: fp32 [nparts, C].   C % 512 == 0, D == 128.
 * eps < 0: the rows are NOT normalised — y = x * w_p[c], then RoPE: the reference's nn.Identity in place of WanRMSNorm when a model is
 *   built with qk_norm=False (wan23/modules/model.py:175-176); w then carries ones (times the attention scale on the q side). The same
 *   holds for yume_rmsnorm_rows_periodic.
 */
int yume_rmsnorm_rope(void* buf, int64_t ld, int64_t T, int64_t C, int nparts,
                      const float* w, float eps, const float* rope, int64_t D, void* stream);
/* same RMSNorm (no RoPE) with a per-row weight: row t uses w[(t % wperiod), :]. Normalises the cross-attention K of ALL
 * blocks (wan23/modules/model.py:222-226, one WanRMSNorm per block) in one launch over the buffer [tokens, wperiod*C]
 * viewed as [tokens*wperiod, C]. w: fp32 [wperiod, C]. */
int yume_rmsnorm_rows_periodic(void* buf, int64_t ld, int64_t T, int64_t C, const float* w, int64_t wperiod, float eps,
                               void* stream);

/* ---- exact-softmax attention forward (FlashAttention-style, head_dim 128) -----------------
 * replaces: wan23/modules/attention.py:24-130 flash_attention() -> flash_attn_varlen_func
 *           (external flash-attn 2.7.0.post2), call sites wan23/modules/model.py:197-202,227.
 *
 *   O[h, i, :] = softmax_j( scale * <Q[h,i,:], K[h,j,:]> ) V[h, j, :]     j < Lk, no mask, no dropout
 * Q: bf16, token-major: element (i, h, d) at Q[i*ldq + h*128 + d];  K likewise (ldk).
 * Vt: bf16 K-major ("V transposed"): element (j, h, d) at Vt[(h*128 + d)*ldvt + j], ldvt >= Lk,
 *     ldvt % 8 == 0 (the image written by YUME_EPI_BF16_SPLITT / yume_transpose_bf16).
 * O: bf16 token-major (ldo). accumulate != 0: O += result (the 14B image cross-attention sum,
 *    wan/modules/model.py:379-387) using the fp32 accumulator before rounding.
 * variant: 0 = automatic (Lk >= 1536 and Lq >= 256: the one-wave-per-SIMD kernel, 256 queries per workgroup; otherwise
 *    the 4-wave LDS-DMA kernel), 1 = 4-wave register-staged kernel, 2 = 4-wave LDS-DMA kernel, 4 = 8-wave ping-pong
 *    kernel, 7 = one-wave-per-SIMD kernel (attn_fwd7.hip), 8 = its persistent form (attn_fwd8.hip; needs the two flags below),
 *    9 = the short-key kernel with K and V^T resident in registers (attn_cross_rk.hpp, r6: 448 < Lk <= 512, Lq >= 1024, ldvt >= 512, and
 *    for Lk < 512 YUME_ATTN_KV_PADDED — the 512-token text cross-attention; measured slower than the 4-wave kernel as built, so variant 0
 *    takes it only with env YUME_ATTN_RK=1). All compute the same function (tests compare them).
 *    | YUME_ATTN_Q_PRESCALED: Q already carries scale * log2(e) — the caller folded that factor into the producer of Q before
 *    its one bf16 rounding (the DiT engine multiplies it into the RMSNorm weight of q, so yume_rmsnorm_rope writes it) — and
 *    `scale` is ignored:  O = sum_j 2^<Q,K_j> V_j / sum_j 2^<Q,K_j>.  The scores then leave the matrix pipe as the exponents
 *    themselves; the one-wave-per-SIMD kernel drops the per-score shift and keeps NO running base (floating point is
 *    scale-invariant: exp2(s - m) and exp2(s) carry the same relative error, m cancels in O / l). Its guard is a range check
 *    of every row sum and output at the end of a workgroup; a workgroup that fails it (scores beyond about +-100 in the log2
 *    domain) is rerun in the same launch on the kernel's rescaling path, so the result is defined for every input.
 */
#define YUME_ATTN_Q_PRESCALED 0x100
/*    | YUME_ATTN_KV_PADDED: the caller guarantees that K has at least ceil(Lk/64)*64 readable rows (whatever they hold) and that Vt has
 *    ldvt >= ceil(Lk/64)*64 with FINITE values in the columns >= Lk (they are multiplied by exact zeros). It opens the persistent
 *    kernel (attn_fwd8.hip: one resident workgroup per CU draws (head, query block) items by ticket and streams K / V^T tiles
 *    continuously across them — the ragged last key tile is fetched like any other and only masked). Which calls take it:
 *      variant 0 (automatic): both flags AND Lk >= 1536 AND Lq >= 256 (where variant 0 would take the one-wave-per-SIMD kernel at all;
 *        the 512-key cross-attention stays on the 4-wave kernel, which measured faster there — also than r6's variant 9) AND a registered counter workspace
 *        (yume_counter_workspace_init) AND Lq * ldq * 2 + 512 < 2^32 (the kernel addresses a query row by a 32-bit byte offset from
 *        its head's base). A call that misses one of these runs variant 7 / 2 as before — silently, it is the same function;
 *        env YUME_ATTN_V8=0 keeps variant 0 off the persistent kernel (A/B runs);
 *      variant 8: insists on it — both flags, Lk >= 512, Lq >= 256, the workspace and the 32-bit limit are then REQUIRED (YUME_EINVAL
 *        by name otherwise). It is the only way to run the persistent kernel for 512 <= Lk < 1536.
 *    Same arithmetic per key tile in the same order as variant 7: whole query blocks are bit-identical. */
#define YUME_ATTN_KV_PADDED 0x200
int yume_attn_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* Vt, int64_t ldvt,
                  void* O, int64_t ldo, int64_t Lq, int64_t Lk, int64_t H, float scale,
                  int accumulate, int variant, void* stream);
/* Same call with caller-owned scratch (yume_attn_workspace_bytes(Lq, Lk, H) bytes, 0 = none needed; 16-byte aligned): in the
 * automatic mode the query blocks that would form a partial last round of workgroups are then cut into 2-4 key ranges inside
 * the same launch (fp32 partial O, running max and row sum in the scratch) and merged in a fixed order. */
int64_t yume_attn_workspace_bytes(int64_t Lq, int64_t Lk, int64_t H);
int yume_attn_fwd_ws(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* Vt, int64_t ldvt,
                     void* O, int64_t ldo, int64_t Lq, int64_t Lk, int64_t H, float scale,
                     int accumulate, int variant, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- small-M fp32 linear (time embedding MLP) ---------------------------------------------
 * replaces: wan23/modules/model.py:459-461,803-812 (time_embedding, time_projection under
 *           autocast(float32)); only R distinct timesteps are evaluated (R <= 8).
 *   out[r, n] = bias[n] + sum_k act(in[r, k]) * W[n, k] ;  act: 0 = identity, 1 = SiLU
 *   out_act: 0 = none, 1 = SiLU applied to the result.
 * W: fp32 (w_bf16 = 0) or bf16 (w_bf16 = 1), [N, K] row-major. K % 8 == 0.
 * add_table (optional, fp32 [N]): out[r, n] += add_table[n] (folds `modulation + e0`).
 */
int yume_linear_smallm_f32(const float* in, int64_t R, int64_t K, const void* W, int w_bf16,
                           const float* bias, int64_t N, int in_act, int out_act,
                           const float* add_table, float* out, void* stream);

/* ---- sinusoidal timestep embedding ---------------------------------------------------------
 * replaces: wan23/modules/model.py:14-24 sinusoidal_embedding_1d (fp64 math, fp32 result).
 * t: fp64 device array (the reference casts positions to float64); row r uses t[t_index[r]]
 * (t_index: device int32 [R], or NULL for t[r]).  out fp32 [R, dim], dim even.
 */
int yume_sinusoidal_embed(const double* t, const int32_t* t_index, int64_t R, int64_t dim,
                          float* out, void* stream);

/* ---- modulation tables -----------------------------------------------------------------------
 * replaces: wan23/modules/model.py:295-297 `(self.modulation.unsqueeze(0) + e).chunk(6)` for every
 *           block at once, and :344 for the head: out[b, r, :] = tab[b, :] + e0[r, :].
 * tab fp32 [B, W] (stacked block.modulation, W = 6*C), e0 fp32 [R, W], out fp32 [B, R, W]. W % 4 == 0.
 */
int yume_modulation_table(const float* tab, const float* e0, int64_t B, int64_t R, int64_t W,
                          float* out, void* stream);

/* ---- patch gather (im2col for stride==kernel Conv3d) ----------------------------------------
 * replaces: wan23/modules/model.py:602-720 patch_embedding{,_2x,_4x,_8x,_16x}(convpadd(...)):
 *           gathers the (1,kh,kw) patches of `nf` frames into a bf16 matrix the GEMM consumes.
 * x: fp32 or bf16 [Cin, F, H, W] contiguous (in_bf16 selects); frames f0 .. f0+nf-1.
 * out: bf16 [nf*Hp*Wp, Kp] with Hp = ceil(H/kh), Wp = ceil(W/kw), column order (c, dh, dw),
 *      zero for h >= H or w >= W (convpadd :918-931) and for columns >= Cin*kh*kw (K padding to Kp).
 */
int yume_patch_gather(const void* x, int in_bf16, int64_t Cin, int64_t F, int64_t H, int64_t W,
                      int64_t f0, int64_t nf, int64_t kh, int64_t kw,
                      void* out, int64_t Kp, void* stream);

/* ---- unpatchify ------------------------------------------------------------------------------
 * replaces: wan23/modules/model.py:867-890 (einsum 'fhwpqrc->cfphqwr', patch (1,ph,pw)).
 * in: fp32 [F*Hp*Wp, ph*pw*Cout] (ld = ldi) ; out: fp32 [Cout, F, Hp*ph, Wp*pw].
 */
int yume_unpatchify(const float* in, int64_t ldi, int64_t Fr, int64_t Hp, int64_t Wp,
                    int64_t ph, int64_t pw, int64_t Cout, float* out, void* stream);

/* ---- layout / dtype helpers -------------------------------------------------------------------
 * yume_cast_bf16: fp32 [rows, cols] (ldi) -> bf16 (ldo), rows beyond `rows_valid` written as 0
 *   (zero-padding the text context to text_len, wan23/modules/model.py:816-821).
 * yume_transpose_bf16: in [rows, cols] (fp32 or bf16, ldi) -> out bf16 [cols, rows] (ldo); builds the
 *   K-major V^T image for callers that enter at the flash_attention() seam with token-major v.
 */
int yume_cast_bf16(const float* in, int64_t ldi, int64_t rows_valid, int64_t rows, int64_t cols,
                   void* out, int64_t ldo, void* stream);
int yume_transpose_bf16(const void* in, int in_bf16, int64_t ldi, int64_t rows, int64_t cols,
                        void* out, int64_t ldo, void* stream);

/* ======================================================================================================
 * Causal 3D VAE (Wan2.2 z=48 stride 4x16x16 and Wan2.1 z=16 stride 4x8x8).
 * Activations are bf16 CHANNELS-LAST [T, H, W, C] (row = one (t,h,w) position, `ldc` elements apart, C % 8 == 0);
 * the reference layout fp32 [C, T, H, W] only exists at the two ends (yume_vae_pack_input / yume_vae_unpack_output).
 * ====================================================================================================== */

/* ---- implicit-GEMM convolution -------------------------------------------------------------------------
 * replaces: wan23/modules/vae2_2.py:17-44 CausalConv3d (3x3x3, (3,1,1), 1x1x1; the 2-frame feat_cache is the
 *           `cache` operand instead of torch.cat + F.pad), :88-98 Upsample(nearest-exact x2)+Conv2d 3x3 (ups=1),
 *           :101-110 ZeroPad2d((0,1,0,1))+Conv2d 3x3 stride 2 and the stride-(2,1,1) time_conv; same classes in
 *           wan/modules/vae.py.
 *   out[(to,ho,wo), co] = bias[co] + sum_{dt,dh,dw,ci} in(ti,hi,wi)[ci] * W[co, ((dt*kh+dh)*kw+dw)*Cin + ci]
 *   ti = to*st + dt - pt: ti < 0 reads cache frame 2+ti (cache = the last two input frames of the previous
 *   chunk, [2,Hin,Win,ldc]) or zeros when cache is NULL; hi = ho*sh + dh - ph, wi likewise, zero outside the frame;
 *   ups != 0: (hi, wi) index a nearest-2x-upsampled view of the input (hi>>1, wi>>1).
 * W: bf16 [Cout, ldw], ldw >= kt*kh*kw*Cin rounded up to 64, zero padded. zero_page: >= 16 zero bytes on the device.
 * Kernel choice (r3, inside the call; every path computes the same sum in fp32, in its own order): stride-1 convolutions whose frames
 *   are whole 256-position tiles with Cin % 64 == 0 (plain or with the folded upsample) run on the one-wave-per-SIMD pipeline
 *   (csrc/conv_w4.hpp; env YUME_CONV_W4=0: the 8-wave kernel); a causal 3x3x3 conv with Cout <= 16 on frames of >= 64 Ki positions
 *   (the decoder head) on the halo-tile kernel (csrc/conv_halo.hpp; YUME_CONV_HALO=0); the 3x3 (x3) stride-1 convolutions of the 96 / 160-channel
 *   levels on csrc/conv_halo_n.hpp (r6; YUME_CONV_HALO_N=0); the encoders' first convolution (8 -> 96, 16 -> 160 channels) on csrc/conv_in.hpp
 *   (r6: weights resident in registers; YUME_CONV_IN=0); everything else on the GEMM kernels with a
 *   gathering A loader. YUME_CONV_KORDER=0/1/2 selects the K walk of that loader (default 2: dt, channel tile, dh, dw).
 * epi: YUME_EPI_BF16 (bias), YUME_EPI_F32, YUME_CONV_EPI_ADD (out = acc + bias + add[m, co], add bf16 [M, ldadd] —
 *      the ResidualBlock skip, vae2_2.py:239), YUME_CONV_EPI_TSPLIT (upsample3d time_conv, vae2_2.py:145-153:
 *      channel halves of frame t become frames 2t and 2t+1: out[((2t+j)*Ho*Wo + hw), c] for co = j*Cout/2 + c),
 *      YUME_CONV_EPI_RMS_SILU (r6): out = SiLU(RMS_norm(acc + bias) * gamma) — the ResidualBlock's second RMS_norm + SiLU behind its first
 *      convolution (wan/modules/vae.py:75-84 RMS_norm = F.normalize over the channels * sqrt(C) * gamma; :190-207 ResidualBlock; vae2_2.py likewise);
 *      `add` then carries the fp32 gamma[Cout] (ldadd ignored). Fused into the epilogue where one workgroup holds a position's whole channel
 *      row (csrc/conv_halo_n.hpp: Cout 96 / 160; the norm sees the fp32 accumulators); on every other kernel choice the call is the plain
 *      convolution followed by yume_vae_rmsnorm_silu in place (the norm sees the bf16 image): the same function to bf16 rounding.
 *      YUME_CONV_FUSE_NORM=0 forces the second form.
 */
enum { YUME_CONV_EPI_ADD = 16, YUME_CONV_EPI_TSPLIT = 17, YUME_CONV_EPI_RMS_SILU = 18 };
int yume_conv3d_cl(const void* x, const void* cache, int64_t ldc, int64_t Tin, int64_t Hin, int64_t Win, int64_t Cin,
                   const void* W, int64_t ldw, const float* bias, int64_t Cout,
                   int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int ups,
                   int64_t To, int64_t Ho, int64_t Wo, int epi, void* out, int64_t ldo,
                   const void* add, int64_t ldadd, const void* zero_page, void* stream);

/* ---- RMS_norm (+SiLU) over channels ---------------------------------------------------------------------
 * replaces: vae2_2.py:47-61 RMS_norm = F.normalize(x, dim=channel) * sqrt(C) * gamma (+ bias), followed by nn.SiLU
 *           in ResidualBlock (:204-208) and the heads (:560-561,676-677).
 *   y[m, c] = act( x[m, c] / max(||x[m, :]||_2, 1e-12) * sqrt(C) * gamma[c] + beta[c] ),  act = SiLU if silu != 0
 * x, y: bf16 [M, C] (ldx, ldy); gamma fp32 [C]; beta fp32 [C] or NULL. fp32 statistics. C % 8 == 0, C <= 4096.
 * y may be x (in place: every kernel form reads a row whole before it writes it). SiLU is x * rcp(1 + exp2(-x log2 e)) in the r6 forms
 * (1 ulp of fp32, below the bf16 rounding of y).
 */
int yume_vae_rmsnorm_silu(const void* x, int64_t ldx, int64_t M, int64_t C, const float* gamma, const float* beta,
                          int silu, void* y, int64_t ldy, void* stream);

/* ---- DupUp3D shortcut, added in place --------------------------------------------------------------------
 * replaces: vae2_2.py:376-418 DupUp3D + the `x_main + x_shortcut` of Up_ResidualBlock (:499-501).
 *   y[(t', h', w'), oc] += x[(t'+toff)/ft, h'/fs, w'/fs][ (oc*ft*fs*fs + a*fs*fs + b*fs + c) / repeats ]
 *   a = (t'+toff)%ft, b = h'%fs, c = w'%fs, repeats = Cout*ft*fs*fs/Cin; toff = ft-1 on the first chunk (the
 *   reference drops the first ft-1 duplicated frames there), else 0.
 * x bf16 [Tin,Hin,Win,Cin] (ldx); y bf16 [To, Hin*fs, Win*fs, Cout] (ldy), in/out.
 */
int yume_vae_dupup_add(const void* x, int64_t ldx, int64_t Tin, int64_t Hin, int64_t Win, int64_t Cin,
                       void* y, int64_t ldy, int64_t To, int64_t Cout, int ft, int fs, int toff, void* stream);

/* ---- AvgDown3D shortcut, added in place --------------------------------------------------------------------
 * replaces: vae2_2.py:322-373 AvgDown3D + the `x + avg_shortcut(x_copy)` of Down_ResidualBlock (:458).
 *   y[(t,h,w), oc] += mean_{g < G} x'[oc*G + g],  x'[ci*ft*fs*fs + a*fs*fs + b*fs + c] = x[(t*ft + a - padt, h*fs+b, w*fs+c), ci]
 *   G = Cin*ft*fs*fs/Cout; padt = (ft - Tin%ft)%ft zero frames in FRONT (frames with index < 0 contribute 0).
 */
int yume_vae_avgdown_add(const void* x, int64_t ldx, int64_t Tin, int64_t Hin, int64_t Win, int64_t Cin,
                         void* y, int64_t ldy, int64_t Cout, int ft, int fs, void* stream);

/* ---- row softmax for the single-head VAE attention -----------------------------------------------------------
 * replaces: the softmax inside F.scaled_dot_product_attention of AttentionBlock (vae2_2.py:272-276); the two
 *           matmuls around it run on yume_gemm_bf16.  P[r, :n] = softmax(scale * S[r, :n]) (bf16), P[r, n:ldp] = 0.
 */
int yume_softmax_rows(const float* S, int64_t lds, int64_t R, int64_t n, float scale, void* P, int64_t ldp, void* stream);

/* ---- layout conversion at the two ends of the VAE --------------------------------------------------------------
 * yume_vae_pack_input: fp32|bf16 [C, T, H, W] -> bf16 channels-last [T, H/ps, W/ps, Cpad] with
 *   v = x[c] * mul[c] + add[c] (mul/add fp32 [C] or NULL: the `z/scale[1] + scale[0]` of decode, vae2_2.py:833-838)
 *   and, for ps = 2, patchify "b c f (h q) (w r) -> b (c r q) f h w" (vae2_2.py:286-302); channels >= C*ps*ps are 0.
 * yume_vae_unpack_output: bf16 channels-last [T, H, W, ldx] (first Cv channels) -> fp32 [Cv/(ps*ps), T, H*ps, W*ps]
 *   with unpatchify (vae2_2.py:305-319), v = (x - sub[c]) * mul[c] (encode's `(mu - mean) * (1/std)`, :821-826)
 *   and clamp to [lo, hi] when lo < hi (decode's .clamp_(-1, 1), :1066).
 */
int yume_vae_pack_input(const void* x, int in_bf16, int64_t C, int64_t T, int64_t H, int64_t W, int ps,
                        const float* mul, const float* add, void* out, int64_t Cpad, void* stream);
int yume_vae_unpack_output(const void* x, int64_t ldx, int64_t T, int64_t H, int64_t W, int64_t Cv, int ps,
                           const float* sub, const float* mul, float lo, float hi, float* out, void* stream);

/* ---- post-decode frame conversion ---------------------------------------------------------------
 * replaces: fastvideo/sample/sample_5b.py:491-500 save_video -> diffusers==0.32.0 VideoProcessor.postprocess_video
 *           (denormalize `(x * 0.5 + 0.5).clamp(0, 1)`, then numpy_to_pil `(x * 255).round().astype("uint8")`) — the
 *           reference does this on the host after a device->host copy of the fp32 video; here it runs on the decoder's
 *           output in HBM and 1 byte per sample crosses PCIe instead of 4.
 * video: fp32 [C, T, H, W] (C <= 4, T*H*W % 4 == 0) -> out: uint8 [T, H, W, C]. Bit-exact (fp32, round half to even).
 */
int yume_frames_u8(const float* video, int64_t C, int64_t T, int64_t H, int64_t W, void* out, void* stream);
/* The web app's own variant of the same step (webapp_single_gpu.py:117-121 `_postprocess_video`, in-tree code):
 *   ((video.clamp(-1, 1) + 1) / 2 * 255).byte()  — the same affine map, but TRUNCATED to uint8 instead of rounded. Same layouts. */
int yume_frames_u8_trunc(const float* video, int64_t C, int64_t T, int64_t H, int64_t W, void* out, void* stream);

/* ======================================================================================================
 * umT5-XXL text encoder (the step before the path; wan/modules/t5.py:440-513 T5EncoderModel,
 * :262-291 T5Encoder, :143-178 T5SelfAttention). It runs on the DiT's GEMM kernels plus four small pieces:
 *   yume_rmsnorm_f32       T5LayerNorm (t5.py:53-67): out bf16 = x * rsqrt(mean(x^2) + eps) * w, x fp32 [T, C]
 *   YUME_EPI_BF16_GEGLU    T5FeedForward's fc1(x) * gelu_tanh(gate(x)) (t5.py:116-141) in ONE GEMM: W = the rows of gate
 *                          and fc1 interleaved (row 2j = gate_j, row 2j+1 = fc1_j), N = 2*dim_ffn, out bf16 [M, N/2]
 *   yume_gemm_bf16_batched the per-head q.k^T and P.v products (t5.py:105-109), `batch` launches per call
 *   yume_softmax_bias_rows P[h,i,:n] = softmax_j(S[h,i,j] + bias[h, j-i+n-1]) (T5 does not scale; bias = the layer's
 *                          relative-position embedding, t5.py:215-259), P[h,i,n:ldp] = 0
 * Padding tokens are not computed at all: their keys are masked to exactly zero weight in the reference (finfo.min bias)
 * and their rows are dropped by T5EncoderModel.__call__ (`u[:v]`), so running on the valid tokens only is identical.
 */
int yume_rmsnorm_f32(const float* x, int64_t ldx, int64_t T, int64_t C, float eps, const float* w, void* out, int64_t ldo,
                     void* stream);
int yume_gemm_bf16_batched(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw, int64_t strideW,
                           int64_t M, int64_t N, int64_t K, int epi, void* out, int64_t ldo, int64_t strideO,
                           int64_t batch, int variant, void* stream);
int yume_softmax_bias_rows(const float* S, int64_t lds, int64_t strideS, int64_t H, int64_t n, const float* bias,
                           int64_t ldb, void* P, int64_t ldp, int64_t strideP, void* stream);
/* split-K for the encoders' small-M GEMMs (M <= 512 tokens against tens of MB of weights — every nn.Linear of t5.py / clip.py):
 * K is cut into `splits` slices computed side by side on the 128x128 kernel (fp32 partials in `workspace`,
 * yume_gemm_splitk_workspace_bytes = splits*M*N*4, caller-owned), then summed in a fixed order with bias + epilogue
 * (YUME_EPI_F32 / RESID without gate / BF16 / BF16_GELU / BF16_GELU_ERF / BF16_GEGLU). K % (splits*64) == 0, N % 8 == 0. */
int64_t yume_gemm_splitk_workspace_bytes(int64_t M, int64_t N, int splits);
int yume_gemm_bf16_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int64_t M, int64_t N,
                          int64_t K, int epi, void* out, int64_t ldo, int splits, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YUME_HIP_H */
