"""`flash_attention` operator seam (reference: wan/modules/attention.py:24-130 == wan23/modules/attention.py).

Same signature and return convention as the reference wrapper around flash-attn's varlen kernel: q,k,v are
[B, L, N, D] tensors of any float dtype on the GPU, output is [B, Lq, N, D] in q's dtype. Rebinding
`wan23.modules.model.flash_attention = yume_amd.attention.flash_attention` in the reference tree is enough to
route the reference model's attention through the gfx950 kernel.
"""
import math

import torch

from . import ops

__all__ = ["flash_attention", "attention"]


def flash_attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None, causal=False,
                    window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, version=None):
    assert dtype in (torch.float16, torch.bfloat16)
    assert q.device.type == "cuda" and q.size(-1) <= 256
    if causal or dropout_p != 0. or tuple(window_size) != (-1, -1):
        raise NotImplementedError("yume_amd.flash_attention: only full (non-causal, no dropout, no window) attention "
                                  "is used by the Yume models and implemented")
    if dtype != torch.bfloat16:
        raise NotImplementedError("yume_amd.flash_attention computes in bf16 (the reference default)")
    b, lq, n, d = q.shape
    lk = k.size(1)
    if k.size(2) != n or v.size(-1) != d:
        raise NotImplementedError("yume_amd.flash_attention: equal q/k/v head counts and head dims only")
    if d > 128:
        raise NotImplementedError(f"yume_amd.flash_attention: head_dim {d} > 128 is not built (the reference accepts <= 256; "
                                  "no Yume model uses it)")
    d_in = d
    if d < 128:
        # zero columns add nothing to q.k and produce zero output columns: exact. (CLIP ViT-H/14 has head_dim 80.)
        pad = lambda t: torch.nn.functional.pad(t, (0, 128 - d_in))
        q, k, v = pad(q), pad(k), pad(v)
        d = 128
    out_dtype = q.dtype
    out = torch.empty((b, lq, n * d), dtype=torch.bfloat16, device=q.device)
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(d_in)
    for i in range(b):
        nq = int(q_lens[i]) if q_lens is not None else lq
        nk = int(k_lens[i]) if k_lens is not None else lk
        qi = q[i, :nq].reshape(nq, n * d)
        if q_scale is not None:
            qi = qi * q_scale
        qi = qi.to(torch.bfloat16).contiguous()
        ki = k[i, :nk].reshape(nk, n * d).to(torch.bfloat16).contiguous()
        vi = v[i, :nk].reshape(nk, n * d)
        if vi.dtype not in (torch.float32, torch.bfloat16):
            vi = vi.float()
        vt = torch.empty((n * d, (nk + 7) // 8 * 8), dtype=torch.bfloat16, device=q.device)
        ops.transpose_bf16(vi.contiguous(), vt)
        ops.attn_fwd(qi, ki, vt, out[i, :nq], nq, nk, n, scale=scale)
        if nq < lq:
            out[i, nq:].zero_()   # flash-attn's varlen packing leaves padded queries out; the reference never reads them
    return out.view(b, lq, n, d)[..., :d_in].type(out_dtype)


def attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None, causal=False,
              window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, fa_version=None):
    """reference attention.py:133-179: same as flash_attention when a flash kernel exists — here it always does."""
    return flash_attention(q, k, v, q_lens, k_lens, dropout_p, softmax_scale, q_scale, causal, window_size,
                           deterministic, dtype, fa_version)
