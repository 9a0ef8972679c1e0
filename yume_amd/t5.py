"""umT5-XXL text encoder on the gfx950 kernels (SURVEY §8(f).3 — the step before the denoise path).

Drop-in for the encoder half of the reference's `wan/modules/t5.py` (identical in `wan23/modules/t5.py`): the same module
tree and parameter names (`token_embedding`, `blocks.N.{norm1,attn.{q,k,v,o},norm2,ffn.{gate.0,fc1,fc2},pos_embedding.
embedding}`, `norm`) so `models_t5_umt5-xxl-enc-bf16.pth` loads with `load_state_dict`, the same `umt5_xxl(encoder_only=
True, ...)` factory and the same `T5EncoderModel(text_len, dtype, device, checkpoint_path, tokenizer_path)` wrapper whose
`__call__(texts, device)` returns the list of per-prompt `[n_tokens, 4096]` embeddings (t5.py:470-513). The modules
only own parameters; the arithmetic runs through the C-ABI (`include/yume_hip.h`, section "umT5-XXL text encoder").

What is done differently, and why it is the same function:
  * padding tokens are not computed: the reference adds finfo.min to their keys' logits (zero softmax weight exactly,
    t5.py:99-108) and T5EncoderModel drops their rows (`u[:v]`, :513); `T5Encoder.forward` here returns zeros there;
  * q|k|v are one GEMM whose epilogue writes V K-major, gate|fc1 one GEMM with the GEGLU epilogue, the residual stream is
    fp32 (the reference keeps bf16 activations), the logits are fp32 until the softmax.
"""
import math

import torch
import torch.nn as nn

from . import _lib, ops

EPI_BF16_GEGLU = 6


def _round_up(a, b):
    return (a + b - 1) // b * b


class GELU(nn.Module):
    def forward(self, x):
        raise RuntimeError("executed by the fused HIP engine via T5Encoder.forward")


class T5LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))


class T5Attention(nn.Module):
    def __init__(self, dim, dim_attn, num_heads, dropout=0.1):
        assert dim_attn % num_heads == 0
        super().__init__()
        self.dim, self.dim_attn, self.num_heads, self.head_dim = dim, dim_attn, num_heads, dim_attn // num_heads
        self.q = nn.Linear(dim, dim_attn, bias=False)
        self.k = nn.Linear(dim, dim_attn, bias=False)
        self.v = nn.Linear(dim, dim_attn, bias=False)
        self.o = nn.Linear(dim_attn, dim, bias=False)


class T5FeedForward(nn.Module):
    def __init__(self, dim, dim_ffn, dropout=0.1):
        super().__init__()
        self.dim, self.dim_ffn = dim, dim_ffn
        self.gate = nn.Sequential(nn.Linear(dim, dim_ffn, bias=False), GELU())
        self.fc1 = nn.Linear(dim, dim_ffn, bias=False)
        self.fc2 = nn.Linear(dim_ffn, dim, bias=False)


class T5RelativeEmbedding(nn.Module):
    def __init__(self, num_buckets, num_heads, bidirectional, max_dist=128):
        super().__init__()
        self.num_buckets, self.num_heads, self.bidirectional, self.max_dist = num_buckets, num_heads, bidirectional, max_dist
        self.embedding = nn.Embedding(num_buckets, num_heads)

    def buckets(self, rel_pos):
        """reference t5.py:236-259 (_relative_position_bucket), integer tensor in -> bucket index tensor out."""
        if self.bidirectional:
            nb = self.num_buckets // 2
            out = (rel_pos > 0).long() * nb
            rel_pos = torch.abs(rel_pos)
        else:
            nb = self.num_buckets
            out = torch.zeros_like(rel_pos)
            rel_pos = -torch.min(rel_pos, torch.zeros_like(rel_pos))
        max_exact = nb // 2
        large = max_exact + (torch.log(rel_pos.float() / max_exact) / math.log(self.max_dist / max_exact)
                             * (nb - max_exact)).long()
        large = torch.min(large, torch.full_like(large, nb - 1))
        return out + torch.where(rel_pos < max_exact, rel_pos, large)


class T5SelfAttention(nn.Module):
    def __init__(self, dim, dim_attn, dim_ffn, num_heads, num_buckets, shared_pos=True, dropout=0.1):
        super().__init__()
        self.shared_pos = shared_pos
        self.norm1 = T5LayerNorm(dim)
        self.attn = T5Attention(dim, dim_attn, num_heads, dropout)
        self.norm2 = T5LayerNorm(dim)
        self.ffn = T5FeedForward(dim, dim_ffn, dropout)
        self.pos_embedding = None if shared_pos else T5RelativeEmbedding(num_buckets, num_heads, bidirectional=True)


class T5Encoder(nn.Module):
    def __init__(self, vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, shared_pos=True, dropout=0.1):
        super().__init__()
        self.dim, self.dim_attn, self.dim_ffn = dim, dim_attn, dim_ffn
        self.num_heads, self.num_layers, self.num_buckets, self.shared_pos = num_heads, num_layers, num_buckets, shared_pos
        self.token_embedding = vocab if isinstance(vocab, nn.Embedding) else nn.Embedding(vocab, dim)
        self.pos_embedding = T5RelativeEmbedding(num_buckets, num_heads, bidirectional=True) if shared_pos else None
        self.blocks = nn.ModuleList([T5SelfAttention(dim, dim_attn, dim_ffn, num_heads, num_buckets, shared_pos, dropout)
                                     for _ in range(num_layers)])
        self.norm = T5LayerNorm(dim)
        self._engine = None

    @property
    def engine(self):
        if self._engine is None:
            self._engine = T5Engine(self)
        return self._engine

    @torch.no_grad()
    def forward(self, ids, mask=None):
        """ids [B, L] int64, mask [B, L] (1 = token). Returns [B, L, dim] in the embedding dtype; rows of padding tokens are 0."""
        if ids.device.type != "cuda":
            raise RuntimeError("yume_amd.t5: ids must be on the device — this path has no CPU fallback")
        B, L = ids.shape
        out = torch.zeros((B, L, self.dim), dtype=self.token_embedding.weight.dtype, device=ids.device)
        for b in range(B):
            n = L if mask is None else int(mask[b].gt(0).sum())
            if mask is not None and n < L and bool(mask[b, n:].gt(0).any()):
                raise RuntimeError("yume_amd.t5: the attention mask must be a prefix mask (tokens first, padding after)")
            if n:
                out[b, :n] = self.engine.encode(ids[b, :n]).to(out.dtype)
        return out


class T5Engine:
    """Packed bf16 weights + workspaces for one T5Encoder; `encode(ids [n]) -> fp32 [n, dim]`."""

    def __init__(self, model):
        self.model = model
        self.P = None
        self._key = None
        self._bufs = {}
        self._bias = {}          # n -> per-layer [H, 2n-1] fp32 relative-position bias tables

    def _buf(self, name, shape, dtype, zero=False):
        b = self._bufs.get(name)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype:
            b = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.dev)
            self._bufs[name] = b
        return b

    def ensure_packed(self):
        m = self.model
        key = tuple((p.data_ptr(), p._version) for p in m.parameters())
        if key == self._key:
            return
        self.dev = m.token_embedding.weight.device
        if self.dev.type != "cuda":
            raise RuntimeError("yume_amd.t5: the encoder must live on the device — this path has no CPU fallback")
        if (m.dim_attn // m.num_heads) % 64 or m.dim % 64 or m.dim_ffn % 64:
            raise RuntimeError("yume_amd.t5: head_dim, dim and dim_ffn must be multiples of 64")
        bf = lambda t: t.detach().to(torch.bfloat16).contiguous()
        f32 = lambda t: t.detach().float().contiguous()
        blocks = []
        for blk in m.blocks:
            a, f = blk.attn, blk.ffn
            geglu = torch.stack([f.gate[0].weight.detach(), f.fc1.weight.detach()], dim=1).reshape(2 * m.dim_ffn, m.dim)
            pe = blk.pos_embedding if blk.pos_embedding is not None else m.pos_embedding
            blocks.append(dict(n1=f32(blk.norm1.weight), n2=f32(blk.norm2.weight),
                               wqkv=bf(torch.cat([a.q.weight, a.k.weight, a.v.weight], dim=0)), wo=bf(a.o.weight),
                               wgeglu=bf(geglu), w2=bf(f.fc2.weight), pos=f32(pe.embedding.weight), pe=pe))
        self.P = dict(blocks=blocks, norm=f32(m.norm.weight))
        self._bias.clear()
        self._key = key

    def _bias_tables(self, n):
        tabs = self._bias.get(n)
        if tabs is None:
            rel = torch.arange(-(n - 1), n, device=self.dev)                       # rel = j - i, slot rel + n - 1
            idx = self.P["blocks"][0]["pe"].buckets(rel)
            tabs = [d["pos"].t()[:, idx].contiguous() for d in self.P["blocks"]]    # t5.py:215-234 as a 1-D table per head
            if len(self._bias) > 8:
                self._bias.clear()
            self._bias[n] = tabs
        return tabs

    @torch.no_grad()
    def encode(self, ids):
        self.ensure_packed()
        m = self.model
        C, Da, H, Dff = m.dim, m.dim_attn, m.num_heads, m.dim_ffn
        hd = Da // H
        n = int(ids.numel())
        npad = _round_up(n, 64)
        if npad > 1024:
            raise RuntimeError(f"yume_amd.t5: {n} tokens exceed the 1024-token limit of the attention rows kernel")
        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream
        x = m.token_embedding.weight[ids.reshape(-1).to(self.dev)].float().contiguous()          # [n, C] fp32 residual stream
        h = self._buf("h", (n, C), torch.bfloat16)
        qk = self._buf(f"qk{npad}", (npad, 2 * Da), torch.bfloat16, zero=True)
        vt = self._buf(f"vt{npad}", (Da, npad), torch.bfloat16, zero=True)
        S = self._buf(f"S{npad}_{n}", (H, n, npad), torch.float32)
        Pm = self._buf(f"P{npad}_{n}", (H, n, npad), torch.bfloat16)
        att = self._buf("att", (n, Da), torch.bfloat16)
        ff = self._buf("ff", (n, Dff), torch.bfloat16)
        tabs = self._bias_tables(n)
        for d, bias in zip(self.P["blocks"], tabs):
            _lib.check(lib.yume_rmsnorm_f32(x.data_ptr(), C, n, C, m.blocks[0].norm1.eps, d["n1"].data_ptr(), h.data_ptr(), C, st),
                       "yume_rmsnorm_f32")
            ops.gemm_bf16(h, d["wqkv"], None, qk[:n], ops.EPI_BF16_SPLITT, out_t=vt, n_split=2 * Da)
            # S[h] = q_h . k_h^T (no scaling): A = q columns of head h, W = the k rows (k is [keys, d] already)
            _lib.check(lib.yume_gemm_bf16_batched(qk.data_ptr(), 2 * Da, hd, qk[:, Da:].data_ptr(), 2 * Da, hd, n, npad, hd,
                                                  ops.EPI_F32, S.data_ptr(), npad, n * npad, H, 0, st), "yume_gemm_bf16_batched")
            _lib.check(lib.yume_softmax_bias_rows(S.data_ptr(), npad, n * npad, H, n, bias.data_ptr(), bias.shape[1],
                                                  Pm.data_ptr(), npad, n * npad, st), "yume_softmax_bias_rows")
            # att[:, head h] = P[h] . v_h : W = the K-major V^T rows of head h
            _lib.check(lib.yume_gemm_bf16_batched(Pm.data_ptr(), npad, n * npad, vt.data_ptr(), npad, hd * npad, n, hd, npad,
                                                  ops.EPI_BF16, att.data_ptr(), Da, hd, H, 0, st), "yume_gemm_bf16_batched")
            ops.gemm_small_m(att, d["wo"], None, x, ops.EPI_RESID)
            _lib.check(lib.yume_rmsnorm_f32(x.data_ptr(), C, n, C, m.blocks[0].norm2.eps, d["n2"].data_ptr(), h.data_ptr(), C, st),
                       "yume_rmsnorm_f32")
            ops.gemm_small_m(h, d["wgeglu"], None, ff, EPI_BF16_GEGLU)          # split-K; falls back to the 256x256 GEGLU kernel
            ops.gemm_small_m(ff, d["w2"], None, x, ops.EPI_RESID)
        _lib.check(lib.yume_rmsnorm_f32(x.data_ptr(), C, n, C, m.norm.eps, self.P["norm"].data_ptr(), h.data_ptr(), C, st),
                   "yume_rmsnorm_f32")
        return h.float()


def umt5_xxl(encoder_only=True, return_tokenizer=False, dtype=torch.float32, device="cpu", **kwargs):
    """reference t5.py:455-467 (+ _t5 :393-452), encoder half only."""
    if not encoder_only or return_tokenizer:
        raise NotImplementedError("yume_amd.t5 provides the encoder (encoder_only=True, return_tokenizer=False)")
    cfg = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32,
               shared_pos=False, dropout=0.1)
    ren = {"vocab_size": "vocab", "encoder_layers": "num_layers"}
    for k, v in kwargs.items():
        if k != "decoder_layers":
            cfg[ren.get(k, k)] = v
    with torch.device(device):
        model = T5Encoder(**cfg)
    return model.to(dtype=dtype, device=device)


def clean_prompt(text, fix_text=None):
    """the `clean='whitespace'` preprocessing of the reference tokenizer wrapper (wan/modules/tokenizers.py:12-22,75-77):
    ftfy.fix_text, html.unescape twice, strip, runs of whitespace -> one blank. ftfy (requirements.txt) is required exactly as in
    the reference: without it mojibake would silently tokenise differently, so its absence is an error, not a fallback.
    `fix_text` replaces ftfy.fix_text (tests; callers that repair their text upstream)."""
    import html
    import re
    if fix_text is None:
        try:
            import ftfy
        except ImportError as e:  # pragma: no cover - depends on the environment
            raise RuntimeError("prompt cleaning needs the `ftfy` package (requirements.txt; the reference's tokenizer wrapper calls "
                               "ftfy.fix_text); install it or pass pre-tokenised ids to encode_ids()") from e
        fix_text = ftfy.fix_text
    text = html.unescape(html.unescape(fix_text(text))).strip()
    return re.sub(r"\s+", " ", text).strip()


class T5EncoderModel:
    """reference t5.py:470-513. The tokenizer (HuggingFace `google/umt5-xxl`, reference tokenizers.py) is host-side glue and is
    only built when `tokenizer_path` points at local tokenizer files; `encode_ids(ids, mask)` takes pre-tokenised input."""

    def __init__(self, text_len, dtype=torch.bfloat16, device="cuda", checkpoint_path=None, tokenizer_path=None, shard_fn=None,
                 model=None):
        self.text_len, self.dtype, self.device = text_len, dtype, device
        self.checkpoint_path, self.tokenizer_path = checkpoint_path, tokenizer_path
        if model is None:
            model = umt5_xxl(encoder_only=True, return_tokenizer=False, dtype=dtype, device=device)
            if checkpoint_path is not None:
                model.load_state_dict(torch.load(checkpoint_path, map_location="cpu"))
        self.model = model.eval().requires_grad_(False).to(device)
        self.tokenizer = None
        if tokenizer_path is not None:
            from transformers import AutoTokenizer
            self.tokenizer = AutoTokenizer.from_pretrained(tokenizer_path)

    def encode_ids(self, ids, mask):
        ids, mask = ids.to(self.device), mask.to(self.device)
        seq_lens = mask.gt(0).sum(dim=1).long()
        context = self.model(ids, mask)
        return [u[:v] for u, v in zip(context, seq_lens)]

    def __call__(self, texts, device=None):
        if self.tokenizer is None:
            raise RuntimeError("T5EncoderModel was built without tokenizer files; use encode_ids(ids, mask)")
        if isinstance(texts, str):
            texts = [texts]
        texts = [clean_prompt(t) for t in texts]            # HuggingfaceTokenizer(clean='whitespace'), t5.py:499-500
        enc = self.tokenizer(texts, return_tensors="pt", padding="max_length", truncation=True, max_length=self.text_len,
                             add_special_tokens=True)
        return self.encode_ids(enc.input_ids, enc.attention_mask)
