"""TEST INFRASTRUCTURE — CPU restatement (plain torch fp32) of the reference's umT5 text encoder, wan/modules/t5.py.
Pinned to the real reference classes imported from /root/reference (tests/test_oracle_t5.py, build container) and to the
golden vectors generated from them (oracle/make_golden_t5.py -> tests/golden/t5_tiny.pt). Only tests/ may import this."""
import math

import torch


def rms_norm(x, w, eps=1e-6):
    """T5LayerNorm, t5.py:53-67."""
    return w * (x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps))


def gelu_tanh(x):
    """t5.py:46-50."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def rel_buckets(lq, lk, num_buckets, max_dist=128):
    """T5RelativeEmbedding (bidirectional), t5.py:225-259: bucket index [lq, lk] of rel = j - i."""
    rel = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < max_exact, rel, large)


def encoder_forward(sd, cfg, ids, mask=None):
    """T5Encoder.forward, t5.py:283-291 with T5SelfAttention :170-176, T5Attention :86-113, T5FeedForward :131-141
    (dropout off). ids [B, L], mask [B, L] or None -> [B, L, dim] fp32."""
    H = cfg["num_heads"]
    x = sd["token_embedding.weight"].float()[ids]                                   # [B, L, C]
    B, L, _ = x.shape
    shared = cfg.get("shared_pos", False)
    for i in range(cfg["num_layers"]):
        pre = f"blocks.{i}."
        emb = sd["pos_embedding.embedding.weight" if shared else pre + "pos_embedding.embedding.weight"].float()
        bias = emb[rel_buckets(L, L, cfg["num_buckets"])].permute(2, 0, 1).unsqueeze(0)          # [1, H, L, L]
        h = rms_norm(x, sd[pre + "norm1.weight"].float())
        q = (h @ sd[pre + "attn.q.weight"].float().t()).view(B, L, H, -1)
        k = (h @ sd[pre + "attn.k.weight"].float().t()).view(B, L, H, -1)
        v = (h @ sd[pre + "attn.v.weight"].float().t()).view(B, L, H, -1)
        ab = bias.expand(B, -1, -1, -1).clone()
        if mask is not None:
            ab.masked_fill_(mask.view(B, 1, 1, -1) == 0, torch.finfo(torch.float32).min)
        a = torch.softmax(torch.einsum("binc,bjnc->bnij", q, k) + ab, dim=-1)                    # no 1/sqrt(d) scaling
        o = torch.einsum("bnij,bjnc->binc", a, v).reshape(B, L, -1)
        x = x + o @ sd[pre + "attn.o.weight"].float().t()
        h = rms_norm(x, sd[pre + "norm2.weight"].float())
        ffh = (h @ sd[pre + "ffn.fc1.weight"].float().t()) * gelu_tanh(h @ sd[pre + "ffn.gate.0.weight"].float().t())
        x = x + ffh @ sd[pre + "ffn.fc2.weight"].float().t()
    return rms_norm(x, sd["norm.weight"].float())
