"""DiT denoise engine: drives the HIP kernels (yume_amd.ops) for one WanModel forward.

The nn.Module classes in yume_amd/wan23/modules/model.py and yume_amd/wan/modules/model.py only hold the
parameters under the reference's state_dict key names and keep the reference call signatures; all compute
goes through this engine:

    pack (once)   q|k|v weights concatenated to one [3C, C] bf16 matrix, cross k|v to [2C, C], patch-embed
                  kernels flattened to GEMM operands, head weight split hi/lo for the fp32-accurate head GEMM,
                  block modulations stacked.
    per clip      FramePack plan, per-token RoPE (cos, sin) table and timestep-row selector, cached on device.
    per step      patch gather+GEMM -> fp32 residual stream; time MLP on the R distinct timesteps only
                  (R = 2 on the FramePack path: clean history and the current sigma) instead of all L tokens;
                  text MLP; per block: adaLN -> fused QKV GEMM (V written K-major) -> RMSNorm+RoPE ->
                  attention -> o-proj GEMM with gate*y+residual epilogue -> norm3 -> cross-attention ->
                  adaLN -> FFN (GELU fused) -> gate/residual epilogue; head; unpatchify.

HBM layout (L tokens, C hidden): residual stream x fp32 [L, C]; GEMM operands bf16 row-major [L, K];
q|k bf16 [L, 2C]; V^T bf16 [C, Lpad] (K-major, per head 128 rows); modulation tables fp32 [blocks, R, 6, C].
"""

import math
import os

import torch

from . import framepack, ops
from .ops import EPI_BF16, EPI_BF16_GELU, EPI_BF16_GELU_ERF, EPI_BF16_SPLITT, EPI_F32, EPI_RESID


def _round_up(a, b):
    return (a + b - 1) // b * b


class DiTEngine:
    def __init__(self, model, family):
        assert family in ("wan23", "wan")
        self.model = model
        self.family = family
        self._packed_key = None
        self._ws = {}
        self._plan_cache = {}
        self.prof = None   # dict kernel-group -> list of (start, end) HIP event pairs when bench.py profiles the timed steps
        self.prof_only = None   # set of group names: bracket only these (an event pair costs a few µs of queue time per call; ~330 calls per step)
        # the text / CLIP embeddings and every block's cross-attention K, V^T depend only on the
        # conditioning, not on the latent or the timestep; with cache_context they are computed once per conditioning
        # tensor (same object, same version) instead of once per denoise step. Off by default = the reference's work.
        self.cache_context = False
        # both families drop the history tokens before unpatchify (wan23/modules/model.py:860, wan/modules/model.py:1003-1005),
        # so in the LAST block the history rows only have to supply K / V to the self-attention: with trim_last_block their queries, o
        # projection, cross-attention and FFN are not computed (5B: 2420 of 9460 rows of one block). The returned velocity is unchanged
        # (GEMM / attention rows are independent). Off by default = the reference's work; bench.py reports it beside the headline.
        self.trim_last_block = os.environ.get("YUME_TRIM_LAST_BLOCK", "0") == "1"
        # kernel selection passed to every GEMM / attention call: 0 = automatic (product setting); tests set (1, 1) to run the
        # whole model on the independent 128x128-tile GEMM and register-staged attention kernels as a cross-check
        self.gemm_variant = 0
        self.attn_variant = 0
        # self-attention q leaves yume_rmsnorm_rope already multiplied by softmax scale * log2(e) (folded into the fp32 RMSNorm weight of q,
        # i.e. before q's one bf16 rounding) and yume_attn_fwd is told so (YUME_ATTN_Q_PRESCALED): the scores are the exponents.
        # YUME_ATTN_PRESCALE=0 keeps the scale inside the attention kernel (A/B and cross-checks).
        self.q_prescale = os.environ.get("YUME_ATTN_PRESCALE", "1") != "0"
        # a yume_amd.ulysses.SequenceParallel splits ONE chain's tokens over the ranks of its group
        self.sp = None
        self._ctx_key = None
        self._ctx_refs = None

    # ------------------------------------------------------------------ profiling (bench.py)
    def _timed(self, name, fn, *a, **k):
        """run one kernel call; with self.prof set, bracket it with HIP events recorded on the launch stream (torch's current
        stream IS the stream ops.* enqueue on)."""
        if self.prof is None or (self.prof_only is not None and name not in self.prof_only):
            return fn(*a, **k)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        r = fn(*a, **k)
        ev[1].record()
        self.prof.setdefault(name, []).append(ev)
        return r

    # ------------------------------------------------------------------ weights
    def _param_key(self):
        # walked once per forward: (storage address, in-place version, dtype) of every parameter — a load_state_dict, an
        # optimizer step or a .to(dtype) re-packs; cheap next to one forward (≈1 k tensors), and cached per forward call
        return (self.q_prescale,) + tuple((p.data_ptr(), p._version, p.dtype) for p in self.model.parameters())

    def _pack(self):
        m = self.model
        dev = m.patch_embedding.weight.device
        if dev.type != "cuda":
            raise RuntimeError("yume_amd WanModel must live on the GPU ('cuda') — there is no CPU path")
        C = m.dim

        def bf(w):
            return w.detach().to(torch.bfloat16).contiguous()

        def f32(w):
            return w.detach().to(torch.float32).contiguous()

        def padk(w2d):
            K = w2d.shape[1]
            Kp = _round_up(K, 64)
            if Kp != K:
                w2d = torch.cat([w2d, w2d.new_zeros(w2d.shape[0], Kp - K)], dim=1)
            return w2d.contiguous()

        P = {}
        # patch embeddings (levels 0..4 + the 2x_f pre-conv); only those attached to the module are packed
        P["pe"] = {}
        for lvl, suf in enumerate(framepack.LEVEL_SUFFIX):
            conv = getattr(m, "patch_embedding" + suf, None)
            if conv is not None:
                P["pe"][lvl] = (padk(bf(conv.weight.flatten(1))), f32(conv.bias))
        conv = getattr(m, "patch_embedding_2x_f", None)
        if conv is not None:
            P["pe"]["2x_f"] = (padk(bf(conv.weight.flatten(1))), f32(conv.bias))
        te, tm, tp = m.text_embedding, m.time_embedding, m.time_projection
        P["text"] = (bf(te[0].weight), f32(te[0].bias), bf(te[2].weight), f32(te[2].bias))
        # fp32 islands keep the parameter's own precision (bf16 if the model was cast, else fp32)
        keep = lambda w: w.detach().contiguous() if w.dtype in (torch.float32, torch.bfloat16) else f32(w)
        P["time"] = (keep(tm[0].weight), f32(tm[0].bias), keep(tm[2].weight), f32(tm[2].bias),
                     keep(tp[1].weight), f32(tp[1].bias))
        if self.family == "wan":
            pr = m.img_emb.proj
            P["img"] = (f32(pr[0].weight), f32(pr[0].bias), bf(pr[1].weight), f32(pr[1].bias), bf(pr[3].weight),
                        f32(pr[3].bias), f32(pr[4].weight), f32(pr[4].bias))
        blocks = []
        qs = math.log2(math.e) / math.sqrt(C // m.num_heads) if self.q_prescale else 1.0
        # qk_norm=False (reference model.py:175-176): norm_q / norm_k are nn.Identity — the RoPE / scale kernel runs with its normalisation off
        # (eps < 0, include/yume_hip.h) and a weight of ones (times the attention scale on the q side)
        self.qk_eps = m.eps if getattr(m.blocks[0], "qk_norm", True) else -1.0

        def nw(mod):
            w = getattr(mod, "weight", None)
            return (w if w is not None else torch.ones(C, device=dev)).double()

        for b in m.blocks:
            sa, ca = b.self_attn, b.cross_attn
            d = {
                "wqkv": bf(torch.cat([sa.q.weight, sa.k.weight, sa.v.weight], dim=0)),
                "bqkv": f32(torch.cat([sa.q.bias, sa.k.bias, sa.v.bias])),
                "nqk": f32(torch.cat([nw(sa.norm_q) * qs, nw(sa.norm_k)])),
                "wo": bf(sa.o.weight), "bo": f32(sa.o.bias),
                "wq_c": bf(ca.q.weight), "bq_c": f32(ca.q.bias), "nq_c": f32(nw(ca.norm_q) * qs),   # (cross q carries the scale too)
                "wo_c": bf(ca.o.weight), "bo_c": f32(ca.o.bias),
                "w1": bf(b.ffn[0].weight), "b1": f32(b.ffn[0].bias),
                "w2": bf(b.ffn[2].weight), "b2": f32(b.ffn[2].bias),
            }
            if getattr(b.norm3, "weight", None) is not None:
                d["n3w"], d["n3b"] = f32(b.norm3.weight), f32(b.norm3.bias)
            blocks.append(d)
        P["blocks"] = blocks
        # cross-attention K / V projections of ALL blocks as one weight: the conditioning tokens are the same for every block,
        # so one [n_ctx, C] x [2*nb*C, C]^T GEMM (K rows of all blocks first, then V rows) replaces nb launches whose 128-wide
        # tiles each took a full K loop's latency; same products, same rounding. +2*nb*C*C bf16 of HBM (5B: 1.1 GB).
        cas = [b.cross_attn for b in m.blocks]
        P["wkv_c"] = bf(torch.cat([ca.k.weight for ca in cas] + [ca.v.weight for ca in cas], dim=0))
        P["bkv_c"] = f32(torch.cat([ca.k.bias for ca in cas] + [ca.v.bias for ca in cas]))
        P["nk_c"] = f32(torch.stack([nw(ca.norm_k) for ca in cas]))
        if self.family == "wan":
            P["wkv_i"] = bf(torch.cat([ca.k_img.weight for ca in cas] + [ca.v_img.weight for ca in cas], dim=0))
            P["bkv_i"] = f32(torch.cat([ca.k_img.bias for ca in cas] + [ca.v_img.bias for ca in cas]))
            P["nk_i"] = f32(torch.stack([nw(ca.norm_k_img) for ca in cas]))
        P["mod_all"] = f32(torch.cat([b.modulation.reshape(1, 6 * C) for b in m.blocks], dim=0))
        P["mod_head"] = f32(m.head.modulation.reshape(2, C))
        wh = m.head.head.weight.detach().float()
        hi = wh.to(torch.bfloat16)
        lo = (wh - hi.float()).to(torch.bfloat16)
        P["whead"] = torch.cat([hi, lo, hi], dim=1).contiguous()   # pairs with A' = [hi | hi | lo]
        P["bhead"] = f32(m.head.head.bias)
        self.P = P
        self.dev = dev

    def ensure_packed(self):
        key = self._param_key()
        if key != self._packed_key:
            self._pack()
            self._packed_key = key

    # ------------------------------------------------------------------ workspaces / per-clip tables
    def _buf(self, name, shape, dtype, zero=False):
        """a workspace tensor, kept across calls while its shape holds. zero: zero-filled when (re)allocated — for buffers whose padding
        (rows / columns no kernel writes) has to stay finite."""
        t = self._ws.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != self.dev:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.dev)
            self._ws[name] = t
        return t

    def _clip_tables(self, key, build):
        v = self._plan_cache.get(key)
        if v is None:
            if len(self._plan_cache) > 64:
                self._plan_cache.clear()
            v = build()
            self._plan_cache[key] = v
        return v

    # ------------------------------------------------------------------ pieces
    def _embed_group(self, u, g, xs_rows):
        """patch-embed frames [g.f0, g.f0+g.nf) of u [Cin,F,H,W] at level g.level into xs_rows (fp32 [ntok, C])."""
        P = self.P["pe"]
        if g.level == 5:
            w, b = P["2x_f"]
            Cin = u.shape[0]
            h4, w4 = -(-u.shape[2] // 4), -(-u.shape[3] // 4)
            a = self._buf("pe_a5", (g.nf * h4 * w4, w.shape[1]), torch.bfloat16)
            ops.patch_gather(u, g.f0, g.nf, 4, 4, a)
            mid = self._buf("pe_mid", (g.nf * h4 * w4, Cin), torch.float32)
            ops.gemm_bf16(a, w, b, mid, EPI_F32, variant=self.gemm_variant)
            # layout change only: token-major [f,h,w,c] -> [c,f,h,w] for the second gather
            u = mid.view(g.nf, h4, w4, Cin).permute(3, 0, 1, 2).contiguous()
            f0, lvl = 0, 4
        else:
            f0, lvl = g.f0, g.level
        if lvl not in P:
            raise RuntimeError(f"patch_embedding{framepack.LEVEL_SUFFIX[lvl]} is needed for this history length "
                               "but is not attached to the model")
        w, b = P[lvl]
        k = 2 << lvl
        a = self._buf(f"pe_a{lvl}_{g.ntok}", (g.ntok, w.shape[1]), torch.bfloat16)
        ops.patch_gather(u, f0, g.nf, k, k, a)
        ops.gemm_bf16(a, w, b, xs_rows, EPI_F32, variant=self.gemm_variant)

    def _time_rows(self, t64, t_index, R):
        """e [R, C], e0 [R, 6C] for the R distinct timesteps (fp32 islands of the reference)."""
        C = self.model.dim
        w0, b0, w2, b2, wp, bp = self.P["time"]
        e = self._buf("t_e", (R, C), torch.float32)
        e0 = self._buf("t_e0", (R, 6 * C), torch.float32)
        for r0 in range(0, R, 8):
            r1 = min(R, r0 + 8)
            n = r1 - r0
            s = self._buf("t_sin", (8, self.model.freq_dim), torch.float32)[:n]
            h = self._buf("t_h", (8, C), torch.float32)[:n]
            idx = t_index[r0:r1] if t_index is not None else None
            tt = t64 if t_index is not None else t64[r0:r1]
            ops.sinusoidal_embed(tt, idx, n, self.model.freq_dim, s)
            ops.linear_smallm_f32(s, w0, b0, h, in_act=0, out_act=1)          # Linear -> SiLU
            ops.linear_smallm_f32(h, w2, b2, e[r0:r1])                        # Linear
            ops.linear_smallm_f32(e[r0:r1], wp, bp, e0[r0:r1], in_act=1)      # SiLU -> Linear
        return e, e0

    def _text_ctx(self, context, out_rows):
        m = self.model
        w0, b0, w2, b2 = self.P["text"]
        n = context.shape[0]
        if n > m.text_len:
            raise RuntimeError(f"context has {n} tokens > text_len {m.text_len}")
        cpad = self._buf("ctx_in", (m.text_len, m.text_dim), torch.bfloat16)
        if n:
            ops.cast_bf16(context.to(device=self.dev, dtype=torch.float32).contiguous(), n, cpad)
        else:
            cpad.zero_()                       # empty prompt: the reference pads [0, text_dim] to text_len zero rows (model.py:816-821)
        hid = self._buf("ctx_hid", (m.text_len, m.dim), torch.bfloat16)
        ops.gemm_bf16(cpad, w0, b0, hid, EPI_BF16_GELU, variant=self.gemm_variant)
        ops.gemm_bf16(hid, w2, b2, out_rows, EPI_BF16, variant=self.gemm_variant)

    def _img_ctx(self, clip_fea, out_rows):
        C = self.model.dim
        lw, lb, w1, b1, w3, b3, l4w, l4b = self.P["img"]
        x = clip_fea.to(device=self.dev, dtype=torch.float32).reshape(-1, clip_fea.shape[-1]).contiguous()
        n, d = x.shape
        a = self._buf("img_a", (n, d), torch.bfloat16)
        ops.adaln_modulate(x, lw, lb, 0, None, False, a, 0, eps=1e-5)         # nn.LayerNorm default eps
        h = self._buf("img_h", (n, d), torch.bfloat16)
        ops.gemm_bf16(a, w1, b1, h, EPI_BF16_GELU_ERF, variant=self.gemm_variant)
        y = self._buf("img_y", (n, C), torch.float32)
        ops.gemm_bf16(h, w3, b3, y, EPI_F32, variant=self.gemm_variant)
        ops.adaln_modulate(y, l4w, l4b, 0, None, False, out_rows, 0, eps=1e-5)

    def _cross_kv(self, tag, ctx_rows, nk, wkv, bkv, nk_w, fresh):
        """K (RMS-normalised) and K-major V^T of all blocks for one conditioning stream: kc [nk, nb*C], vct [nb*C, nk8].
        Recomputed only when `fresh` (always, unless cache_context found the same conditioning tensors)."""
        C, nb = self.model.dim, len(self.P["blocks"])
        # rows / columns up to a whole number of 64-key tiles exist and hold zeros: the attention kernels may fetch the ragged last key tile
        # like any other (YUME_ATTN_KV_PADDED); the GEMM below writes nk rows / columns only
        nk64 = _round_up(nk, 64)
        kc = self._buf(f"kc_{tag}_{nk}", (nk64, nb * C), torch.bfloat16, zero=True)[:nk]
        vct = self._buf(f"vct_{tag}_{nk}", (nb * C, nk64), torch.bfloat16, zero=True)
        if fresh:
            ops.gemm_bf16(ctx_rows, wkv, bkv, kc, EPI_BF16_SPLITT, out_t=vct, n_split=nb * C, variant=self.gemm_variant)
            ops.rmsnorm_rows_periodic(kc.view(nk * nb, C), C, nk_w, self.qk_eps)
        return kc, vct

    def _blocks(self, xs, L, tab, row_idx, R, rope, n_rope, ctx, n_img, ctx_fresh=True, n_keys=None, only=None, cache=None, n_trim=0):
        """xs fp32 [L, C] in/out. tab fp32 [nb, R, 6, C]; rope fp32 [n_rope, 64, 2] (n_rope == L here).
        n_trim: the first n_trim rows of the LAST block's output are not needed by the caller (history tokens in front of the head): that
        block computes K / V for every row but everything behind the QKV projection for the rows [n_trim, L) only; xs[:n_trim] is then stale.
        With sequence parallelism L is this rank's (padded) chunk and n_keys the true global token count.
        only: run just these block indices (tab[j] then belongs to only[j]) — the WanAttentionBlock.forward seam.
        cache: (mode, cache_list, tensors) block-residual cache of wan/modules/model.py:985-1000: mode 'record' appends
        bf16 (x_out - x_in) of the listed blocks, 'replay' adds the stored residual instead of running them."""
        m = self.model
        C, H, Fd, eps = m.dim, m.num_heads, m.ffn_dim, m.eps
        # q|k rows and V^T columns exist (zeros) up to a whole number of 64-key tiles: YUME_ATTN_KV_PADDED (see _cross_kv)
        Lp = _round_up(L, 64)
        h = self._buf("h", (L, C), torch.bfloat16)
        qk = self._buf("qk", (Lp, 2 * C), torch.bfloat16, zero=True)[:L]
        vt = self._buf("vt", (C, Lp), torch.bfloat16, zero=True)
        att = self._buf("att", (L, C), torch.bfloat16)
        ff = self._buf("ff", (L, Fd), torch.bfloat16)
        ts = 6 * C  # table row stride
        ntxt = ctx.shape[0] - n_img
        kc_t, vct_t = self._cross_kv("t", ctx[n_img:], ntxt, self.P["wkv_c"], self.P["bkv_c"], self.P["nk_c"], ctx_fresh)
        if n_img:
            kc_i, vct_i = self._cross_kv("i", ctx[:n_img], n_img, self.P["wkv_i"], self.P["bkv_i"], self.P["nk_i"], ctx_fresh)
        T = self._timed
        ids = range(len(self.P["blocks"])) if only is None else only
        for j, i in enumerate(ids):
            d = self.P["blocks"][i]
            if cache is not None and i in cache[1]:
                if cache[0] == "replay":
                    # reference: `x = x + cache[cache_list.index(cnt_blocks - 1)]` (fp32 + bf16), the block is not run
                    xs.add_(cache[2][cache[1].index(i)].reshape(L, C).to(torch.float32))
                    continue
                x_in = xs.clone()
            tb = tab[j if only is not None else i]                      # [R, 6, C]
            shift_sa, scale_sa, gate_sa = tb[:, 0], tb[:, 1], tb[:, 2]
            shift_ff, scale_ff, gate_ff = tb[:, 3], tb[:, 4], tb[:, 5]
            s = n_trim if j == len(ids) - 1 else 0                      # rows [s, L) are the ones whose output is wanted
            if s:
                self._trimmed_block(xs, L, s, d, i, tb, ts, row_idx, rope, eps, h, qk, vt, att, ff, kc_t, vct_t, ntxt,
                                    (kc_i, vct_i, n_img) if n_img else None)
                continue
            # --- self attention
            T("adaln", ops.adaln_modulate, xs, scale_sa, shift_sa, ts, row_idx, True, h, 0, eps)
            T("gemm_qkv", ops.gemm_bf16, h, d["wqkv"], d["bqkv"], qk, EPI_BF16_SPLITT, out_t=vt, n_split=2 * C, variant=self.gemm_variant)
            if n_rope == L:
                T("rmsnorm_rope", ops.rmsnorm_rope, qk, C, 2, d["nqk"], self.qk_eps, rope)
            elif n_rope == 0:
                ops.rmsnorm_rope(qk, C, 2, d["nqk"], self.qk_eps, None)
            else:
                ops.rmsnorm_rope(qk[:n_rope], C, 2, d["nqk"], self.qk_eps, rope)
                ops.rmsnorm_rope(qk[n_rope:], C, 2, d["nqk"], self.qk_eps, None)
            if self.sp is None:
                T("attn_self", ops.attn_fwd, qk[:, :C], qk[:, C:], vt, att, L, n_keys if n_keys is not None else L, H, variant=self.attn_variant,
                  q_prescaled=self.q_prescale, kv_padded=True)
                sa = att
            else:      # Ulysses: all tokens x this rank's heads, then back (2 collectives, yume_amd/ulysses.py)
                qf, kf, vtf = self.sp.exchange_qkv(qk, vt, C)
                of = self._buf("att_sp", (qf.shape[0], qf.shape[1]), torch.bfloat16)
                ops.attn_fwd(qf, kf, vtf, of, qf.shape[0], n_keys, H // self.sp.world, variant=self.attn_variant, q_prescaled=self.q_prescale)
                sa = self.sp.exchange_out(of)
            T("gemm_o", ops.gemm_bf16, sa, d["wo"], d["bo"], xs, EPI_RESID, gate=gate_sa, gate_stride=ts, row_idx=row_idx, variant=self.gemm_variant)
            # --- cross attention
            if "n3w" in d:
                T("adaln", ops.adaln_modulate, xs, d["n3w"], d["n3b"], 0, None, False, h, 0, eps)
            else:
                ops.cast_bf16(xs, L, h)
            T("gemm_cross_q", ops.gemm_bf16, h, d["wq_c"], d["bq_c"], qk[:, :C], EPI_BF16, variant=self.gemm_variant)
            T("rmsnorm_rope", ops.rmsnorm_rope, qk[:, :C], C, 1, d["nq_c"], self.qk_eps)
            T("attn_cross", ops.attn_fwd, qk[:, :C], kc_t[:, i * C:(i + 1) * C], vct_t[i * C:(i + 1) * C], att, L, ntxt, H, variant=self.attn_variant,
              q_prescaled=self.q_prescale, kv_padded=True)
            if n_img:
                T("attn_cross", ops.attn_fwd, qk[:, :C], kc_i[:, i * C:(i + 1) * C], vct_i[i * C:(i + 1) * C], att, L, n_img, H, accumulate=True,
                  variant=self.attn_variant, q_prescaled=self.q_prescale, kv_padded=True)
            T("gemm_cross_o", ops.gemm_bf16, att, d["wo_c"], d["bo_c"], xs, EPI_RESID, variant=self.gemm_variant)
            # --- FFN
            T("adaln", ops.adaln_modulate, xs, scale_ff, shift_ff, ts, row_idx, True, h, 0, eps)
            T("gemm_ffn0", ops.gemm_bf16, h, d["w1"], d["b1"], ff, EPI_BF16_GELU, variant=self.gemm_variant)
            T("gemm_ffn2", ops.gemm_bf16, ff, d["w2"], d["b2"], xs, EPI_RESID, gate=gate_ff, gate_stride=ts, row_idx=row_idx, variant=self.gemm_variant)
            if cache is not None and cache[0] == "record" and i in cache[1]:
                cache[2].append((xs - x_in).to(torch.bfloat16).unsqueeze(0))     # reference keeps [B, L, C] bf16

    def _trimmed_block(self, xs, L, s, d, i, tb, ts, row_idx, rope, eps, h, qk, vt, att, ff, kc_t, vct_t, ntxt, img):
        """the block of _blocks with everything behind the QKV projection restricted to the rows [s, L) (trim_last_block): the same
        kernel calls on row-offset views, so the rows that are computed go through the same arithmetic."""
        m = self.model
        C, H = m.dim, m.num_heads
        T = self._timed
        shift_sa, scale_sa, gate_sa = tb[:, 0], tb[:, 1], tb[:, 2]
        shift_ff, scale_ff, gate_ff = tb[:, 3], tb[:, 4], tb[:, 5]
        xo, ho, ao, fo, qo = xs[s:], h[s:], att[s:], ff[s:], qk[s:, :C]
        ro = row_idx[s:] if row_idx is not None else None
        n = L - s
        T("adaln", ops.adaln_modulate, xs, scale_sa, shift_sa, ts, row_idx, True, h, 0, eps)
        T("gemm_qkv", ops.gemm_bf16, h, d["wqkv"], d["bqkv"], qk, EPI_BF16_SPLITT, out_t=vt, n_split=2 * C, variant=self.gemm_variant)
        T("rmsnorm_rope", ops.rmsnorm_rope, qk, C, 2, d["nqk"], self.qk_eps, rope)
        T("attn_self", ops.attn_fwd, qo, qk[:, C:], vt, ao, n, L, H, variant=self.attn_variant, q_prescaled=self.q_prescale, kv_padded=True)
        T("gemm_o", ops.gemm_bf16, ao, d["wo"], d["bo"], xo, EPI_RESID, gate=gate_sa, gate_stride=ts, row_idx=ro, variant=self.gemm_variant)
        if "n3w" in d:
            T("adaln", ops.adaln_modulate, xo, d["n3w"], d["n3b"], 0, None, False, ho, 0, eps)
        else:
            ops.cast_bf16(xo, n, ho)
        T("gemm_cross_q", ops.gemm_bf16, ho, d["wq_c"], d["bq_c"], qo, EPI_BF16, variant=self.gemm_variant)
        T("rmsnorm_rope", ops.rmsnorm_rope, qo, C, 1, d["nq_c"], self.qk_eps)
        T("attn_cross", ops.attn_fwd, qo, kc_t[:, i * C:(i + 1) * C], vct_t[i * C:(i + 1) * C], ao, n, ntxt, H, variant=self.attn_variant,
          q_prescaled=self.q_prescale, kv_padded=True)
        if img is not None:
            kc_i, vct_i, n_img = img
            T("attn_cross", ops.attn_fwd, qo, kc_i[:, i * C:(i + 1) * C], vct_i[i * C:(i + 1) * C], ao, n, n_img, H, accumulate=True,
              variant=self.attn_variant, q_prescaled=self.q_prescale, kv_padded=True)
        T("gemm_cross_o", ops.gemm_bf16, ao, d["wo_c"], d["bo_c"], xo, EPI_RESID, variant=self.gemm_variant)
        T("adaln", ops.adaln_modulate, xo, scale_ff, shift_ff, ts, ro, True, ho, 0, eps)
        T("gemm_ffn0", ops.gemm_bf16, ho, d["w1"], d["b1"], fo, EPI_BF16_GELU, variant=self.gemm_variant)
        T("gemm_ffn2", ops.gemm_bf16, fo, d["w2"], d["b2"], xo, EPI_RESID, gate=gate_ff, gate_stride=ts, row_idx=ro, variant=self.gemm_variant)

    # ------------------------------------------------------------------ one block through the engine (operator seam)
    @torch.no_grad()
    def block_forward(self, i, x, e, rope_cs, context, n_img=0):
        """WanAttentionBlock.forward for block i (reference wan23/modules/model.py:272-316, wan/modules/model.py:444-493):
        x [L, C] float; e fp32 [L, 6, C] (per token, 5B) or [1|6, C] (one row, 14B) — the time projection BEFORE the
        block's own modulation is added; rope_cs fp32 [n_rope, 64, 2] (cos, sin) of the first n_rope tokens or None;
        context [Lc, C] already embedded (the first n_img rows are CLIP image tokens). Returns fp32 [L, C]."""
        self.ensure_packed()
        C = self.model.dim
        L = x.shape[0]
        xs = self._buf("xs_blk", (L, C), torch.float32)
        xs.copy_(x.to(device=self.dev, dtype=torch.float32))
        e2 = e.to(device=self.dev, dtype=torch.float32).reshape(-1, 6 * C).contiguous()
        R = e2.shape[0]
        if R not in (1, L):
            raise RuntimeError(f"block modulation e has {R} rows for {L} tokens (expected 1 or one per token)")
        tab = self._buf("tab_blk", (1, R, 6 * C), torch.float32)
        ops.modulation_table(self.P["mod_all"][i:i + 1].contiguous(), e2, tab)
        row_idx = torch.arange(L, dtype=torch.int32, device=self.dev) if R > 1 else None
        ctx = context.to(device=self.dev, dtype=torch.bfloat16).contiguous()
        n_rope = 0 if rope_cs is None else rope_cs.shape[0]
        if n_rope == 0:
            rope_cs = torch.zeros((1, 64, 2), dtype=torch.float32, device=self.dev)
        self._ctx_key = None                                     # the per-conditioning cache belongs to forward_one
        self._blocks(xs, L, tab.view(1, R, 6, C), row_idx, R, rope_cs.to(self.dev).contiguous(), n_rope, ctx, n_img, True, only=[i])
        return xs.clone()

    def _head(self, xs_new, row_idx_new, e, R, grid):
        """xs_new fp32 [Ln, C] -> fp32 [Cout, F, 2*Hp, 2*Wp]."""
        return self._unpatchify(self._head_rows(xs_new, row_idx_new, e, R), grid)

    def _unpatchify(self, y, grid):
        Co = self.model.out_dim
        Fr, Hp, Wp = grid
        out = torch.empty((Co, Fr, 2 * Hp, 2 * Wp), dtype=torch.float32, device=self.dev)
        ops.unpatchify(y, Fr, Hp, Wp, 2, 2, Co, out)
        return out

    def _head_rows(self, xs_new, row_idx_new, e, R):
        """xs_new fp32 [Ln, C] -> head output rows fp32 [Ln, 4*Cout] (model.py:344-347)."""
        m = self.model
        C, Co = m.dim, m.out_dim
        Ln = xs_new.shape[0]
        # (head.modulation [2, C] + e [R, C]) -> th[j, r, :]: j = 0 shift, 1 scale   (model.py:344)
        th = self._buf("tab_head", (2, R, C), torch.float32)
        ops.modulation_table(self.P["mod_head"], e, th)
        a3 = self._buf("head_a", (Ln, 3 * C), torch.bfloat16)
        ops.adaln_modulate(xs_new, th[1], th[0], C, row_idx_new, True, a3, 2, m.eps)
        y = self._buf("head_y", (Ln, 4 * Co), torch.float32)
        ops.gemm_bf16(a3, self.P["whead"], self.P["bhead"], y, EPI_F32, variant=self.gemm_variant)
        return y

    # ------------------------------------------------------------------ one sample forward
    @torch.no_grad()
    def forward_one(self, u, t, context, clip_fea=None, packed=True, lfz=8, n_sel=None, cache=None):
        """u [Cin, F, H, W] (fp32|bf16, x and y already concatenated); t tensor; context [Ltxt, text_dim].
        Returns fp32 [Cout, F', H, W]."""
        self.ensure_packed()
        m = self.model
        C, D = m.dim, m.dim // m.num_heads
        if D != 128:
            raise RuntimeError("yume_amd attention kernels are built for head_dim 128")
        u = u.to(self.dev)
        if u.dtype not in (torch.float32, torch.bfloat16):
            u = u.float()
        u = u.contiguous()
        Cin, F, H, W = u.shape
        if Cin != m.in_dim:
            raise RuntimeError(f"input has {Cin} channels, model in_dim is {m.in_dim}")
        t64 = t.to(device=self.dev, dtype=torch.float64).reshape(-1).contiguous()

        if packed:
            def build():
                plan = framepack.pack_plan(F, H, W, lfz, n_sel)
                rope = framepack.plan_rope(plan, D).to(self.dev)
                ridx = torch.cat([torch.zeros(plan.n_hist_tok, dtype=torch.int32),
                                  torch.ones(plan.n_new_tok, dtype=torch.int32)]).to(self.dev)
                return plan, rope, ridx
            plan, rope, ridx = self._clip_tables(("p", F, H, W, lfz, n_sel), build)
            L, n_hist = plan.seq_len, plan.n_hist_tok
            groups, grid = plan.groups, plan.new_grid
        else:
            def build():
                if H % 2 or W % 2:
                    # the base patch_embedding is an UNPADDED stride-2 Conv3d (model.py:455): odd sizes lose their last row / column
                    # there, while the pyramid levels pad (convpadd); no shipped resolution is odd — refuse instead of guessing
                    raise RuntimeError(f"latent H x W = {H} x {W} must be even for the (1, 2, 2) patch embedding")
                hp, wp = H // 2, W // 2
                rope = framepack.rope_cos_sin([(0, F, hp, wp)], D).to(self.dev)
                return framepack.Group(0, F, 0, hp, wp), rope
            g0, rope = self._clip_tables(("u", F, H, W), build)
            L, n_hist = g0.ntok, 0
            groups, grid = [g0], (F, g0.hp, g0.wp)
            ridx = None

        # --- timestep rows
        if self.family == "wan":
            if t64.numel() != 1:
                raise RuntimeError("the 14B-arch model takes one scalar timestep per sample")
            R, t_index, row_idx = 1, None, None
        elif packed:
            if t64.numel() < 2:
                raise RuntimeError("FramePack path expects a per-token timestep vector (t[0] history, t[-1] new)")
            t_index = self._clip_tables(("ti", t64.numel()), lambda: torch.tensor(
                [0, t64.numel() - 1], dtype=torch.int32, device=self.dev))
            R, row_idx = 2, ridx
        elif t64.numel() == 1:
            R, t_index, row_idx = 1, None, None
        else:
            # arbitrary per-token timesteps (plain path): evaluate the distinct values only
            if t64.numel() < L:
                raise RuntimeError(f"per-token t has {t64.numel()} entries for {L} tokens")
            uniq, inv = torch.unique(t64[:L], return_inverse=True)
            t64, t_index, R = uniq.contiguous(), None, int(uniq.numel())
            row_idx = inv.to(torch.int32).contiguous()

        # --- embeddings
        xs = self._buf("xs", (L, C), torch.float32)
        off = 0
        for g in groups:
            self._embed_group(u, g, xs[off:off + g.ntok])
            off += g.ntok
        e, e0 = self._time_rows(t64, t_index, R)
        nb = len(self.P["blocks"])
        tab = self._buf("tab", (nb, R, 6 * C), torch.float32)
        ops.modulation_table(self.P["mod_all"], e0, tab)
        n_img = 0
        if self.family == "wan":
            if clip_fea is None:
                raise RuntimeError("clip_fea is required by the i2v model")
            n_img = clip_fea.reshape(-1, clip_fea.shape[-1]).shape[0]
        ctx = self._buf("ctx", (n_img + m.text_len, C), torch.bfloat16)
        ctx_fresh = True
        if self.cache_context:
            # storage address + version counter + shape (views of one tensor share all three); holding the references below
            # keeps those addresses from being handed to another tensor while the entry is live
            key = (context.data_ptr(), context._version, tuple(context.shape),
                   None if clip_fea is None else (clip_fea.data_ptr(), clip_fea._version, tuple(clip_fea.shape)),
                   self._packed_key)
            ctx_fresh = key != self._ctx_key
            self._ctx_key, self._ctx_refs = key, (context, clip_fea)
        else:
            self._ctx_key = None
        if ctx_fresh:
            if n_img:
                self._img_ctx(clip_fea, ctx[:n_img])
            self._text_ctx(context, ctx[n_img:])

        if self.sp is not None:
            if cache is not None:
                raise NotImplementedError("cache_sample is not combined with sequence parallelism")
            return self._forward_sp(xs, L, n_hist, tab.view(nb, R, 6, C), row_idx, R, rope, ctx, n_img, ctx_fresh, e, grid)
        n_trim = n_hist if (self.trim_last_block and cache is None) else 0
        self._blocks(xs, L, tab.view(nb, R, 6, C), row_idx, R, rope, L, ctx, n_img, ctx_fresh, cache=cache, n_trim=n_trim)
        ridx_new = row_idx[n_hist:] if row_idx is not None else None
        return self._head(xs[n_hist:], ridx_new, e, R, grid)

    def _forward_sp(self, xs, L, n_hist, tab, row_idx, R, rope, ctx, n_img, ctx_fresh, e, grid):
        """Blocks + head on this rank's token chunk (sequence_parallel.py:121-152: chunk after the embeddings, gather
        after the head). The embeddings above were computed for all L tokens on every rank (< 0.1 % of the FLOPs)."""
        sp, C = self.sp, self.model.dim
        sp.check_heads(self.model.num_heads)
        Lp, lo, hi = sp.chunk(L)
        n = hi - lo
        xl = self._buf("xs_sp", (Lp, C), torch.float32)
        xl[:n].copy_(xs[lo:hi])
        xl[n:].zero_()                                   # pad tokens: finite rows, masked as keys, dropped at the end
        rl = self._buf("rope_sp", (Lp,) + tuple(rope.shape[1:]), torch.float32)
        rl[:n].copy_(rope[lo:hi])
        rl[n:].zero_()
        il = None
        if row_idx is not None:
            il = self._buf("ridx_sp", (Lp,), torch.int32)
            il[:n].copy_(row_idx[lo:hi])
            il[n:].zero_()
        self._blocks(xl, Lp, tab, il, R, rl, Lp, ctx, n_img, ctx_fresh, n_keys=L)
        y = sp.gather_rows(self._head_rows(xl, il, e, R))
        return self._unpatchify(y[n_hist:L].contiguous(), grid)
