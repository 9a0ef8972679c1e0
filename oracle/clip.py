"""TEST INFRASTRUCTURE — CPU restatement (plain torch fp32) of the reference's CLIP vision tower, wan/modules/clip.py
VisionTransformer.forward (:279-300) with AttentionBlock (:146-153), SelfAttention (:74-91), LayerNorm (:47-50).
Pinned to the real class (tests/test_oracle_clip.py) and to tests/golden/clip_tiny.pt. Only tests/ may import this."""
import torch
import torch.nn.functional as F


def layer_norm(x, w, b, eps):
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps)


def visual_forward(sd, cfg, x, use_31_block=True):
    """x [B, 3, S, S] -> [B, 1 + patches, dim]."""
    C, H, eps = cfg["dim"], cfg["num_heads"], cfg["norm_eps"]
    B = x.shape[0]
    t = F.conv2d(x.float(), sd["patch_embedding.weight"].float(), stride=cfg["patch_size"]).flatten(2).permute(0, 2, 1)
    t = torch.cat([sd["cls_embedding"].float().expand(B, -1, -1), t], dim=1) + sd["pos_embedding"].float()
    t = layer_norm(t, sd["pre_norm.weight"], sd["pre_norm.bias"], eps)
    L = t.shape[1]
    n = cfg["num_layers"] - 1 if use_31_block else cfg["num_layers"]
    for i in range(n):
        p = f"transformer.{i}."
        h = layer_norm(t, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        qkv = (h @ sd[p + "attn.to_qkv.weight"].float().t() + sd[p + "attn.to_qkv.bias"].float()).view(B, L, 3, H, C // H)
        q, k, v = (u.transpose(1, 2) for u in qkv.unbind(2))                       # [B, H, L, d]
        a = torch.softmax(q @ k.transpose(-1, -2) * (C // H) ** -0.5, dim=-1) @ v   # flash_attention default scale
        t = t + a.transpose(1, 2).reshape(B, L, C) @ sd[p + "attn.proj.weight"].float().t() + sd[p + "attn.proj.bias"].float()
        h = layer_norm(t, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        h = F.gelu(h @ sd[p + "mlp.0.weight"].float().t() + sd[p + "mlp.0.bias"].float())
        t = t + h @ sd[p + "mlp.2.weight"].float().t() + sd[p + "mlp.2.bias"].float()
    return t
