"""CLIP ViT-H/14 vision tower on the gfx950 kernels (SURVEY §8(f).3 — produces the `clip_fea` [257, 1280] the 14B i2v model
conditions on; reference wan/modules/clip.py:209-300 VisionTransformer, :501-542 CLIPModel.visual).

Drop-in surface: `VisionTransformer` with the reference's constructor arguments and parameter names (`patch_embedding`,
`cls_embedding`, `pos_embedding`, `pre_norm`, `transformer.N.{norm1,attn.{to_qkv,proj},norm2,mlp.{0,2}}`, `post_norm`,
`head`) so the `visual.*` keys of `models_clip_open-clip-xlm-roberta-large-vit-huge-14.pth` load unchanged, and
`CLIPModel(dtype, device, checkpoint_path, tokenizer_path).visual(videos)` (the text tower of that checkpoint is never
run by the Yume pipelines and is not built here). The modules only own parameters; the arithmetic runs through the C-ABI.

The 16 heads of width 80 use the head_dim-128 attention kernel: the fused QKV weight is packed with each head's 80 rows
followed by 48 zero rows (zero q/k columns add nothing to a dot product, zero v rows give zero output columns that meet zero
columns of the packed output projection), the softmax scale is passed explicitly as 80^-1/2. V leaves the QKV GEMM K-major.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

_MEAN = [0.48145466, 0.4578275, 0.40821073]
_STD = [0.26862954, 0.26130258, 0.27577711]


def _round_up(a, b):
    return (a + b - 1) // b * b


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        raise RuntimeError("executed by the fused HIP engine via VisionTransformer.forward")


class SelfAttention(nn.Module):
    def __init__(self, dim, num_heads, causal=False, attn_dropout=0.0, proj_dropout=0.0):
        assert dim % num_heads == 0
        super().__init__()
        self.dim, self.num_heads, self.head_dim, self.causal = dim, num_heads, dim // num_heads, causal
        self.to_qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)


class AttentionBlock(nn.Module):
    def __init__(self, dim, mlp_ratio, num_heads, post_norm=False, causal=False, activation="quick_gelu", attn_dropout=0.0,
                 proj_dropout=0.0, norm_eps=1e-5):
        super().__init__()
        if activation != "gelu" or post_norm or causal:
            raise NotImplementedError("yume_amd.clip implements the ViT-H/14 configuration: pre-norm blocks with nn.GELU")
        self.dim, self.mlp_ratio, self.num_heads, self.norm_eps = dim, mlp_ratio, num_heads, norm_eps
        self.norm1 = LayerNorm(dim, eps=norm_eps)
        self.attn = SelfAttention(dim, num_heads, causal, attn_dropout, proj_dropout)
        self.norm2 = LayerNorm(dim, eps=norm_eps)
        self.mlp = nn.Sequential(nn.Linear(dim, int(dim * mlp_ratio)), nn.GELU(), nn.Linear(int(dim * mlp_ratio), dim),
                                 nn.Dropout(proj_dropout))


class VisionTransformer(nn.Module):
    def __init__(self, image_size=224, patch_size=16, dim=768, mlp_ratio=4, out_dim=512, num_heads=12, num_layers=12,
                 pool_type="token", pre_norm=True, post_norm=False, activation="quick_gelu", attn_dropout=0.0, proj_dropout=0.0,
                 embedding_dropout=0.0, norm_eps=1e-5):
        super().__init__()
        if pool_type != "token" or not pre_norm:
            raise NotImplementedError("yume_amd.clip implements the ViT-H/14 configuration: token pooling, pre_norm")
        self.image_size, self.patch_size, self.num_patches = image_size, patch_size, (image_size // patch_size) ** 2
        self.dim, self.mlp_ratio, self.out_dim, self.num_heads, self.num_layers = dim, mlp_ratio, out_dim or dim, num_heads, num_layers
        self.norm_eps = norm_eps
        gain = 1.0 / math.sqrt(dim)
        self.patch_embedding = nn.Conv2d(3, dim, kernel_size=patch_size, stride=patch_size, bias=False)
        self.cls_embedding = nn.Parameter(gain * torch.randn(1, 1, dim))
        self.pos_embedding = nn.Parameter(gain * torch.randn(1, self.num_patches + 1, dim))
        self.pre_norm = LayerNorm(dim, eps=norm_eps)
        self.transformer = nn.Sequential(*[AttentionBlock(dim, mlp_ratio, num_heads, post_norm, False, activation, attn_dropout,
                                                          proj_dropout, norm_eps) for _ in range(num_layers)])
        self.post_norm = LayerNorm(dim, eps=norm_eps)
        self.head = nn.Parameter(gain * torch.randn(dim, self.out_dim))
        self._engine = None

    @property
    def engine(self):
        if self._engine is None:
            self._engine = ClipEngine(self)
        return self._engine

    @torch.no_grad()
    def forward(self, x, interpolation=False, use_31_block=False):
        """x [B, 3, S, S] normalised images on the device -> [B, 1 + patches, dim] (reference clip.py:279-300)."""
        if interpolation:
            raise NotImplementedError("positional-embedding interpolation is not used by the Yume pipelines")
        if x.device.type != "cuda":
            raise RuntimeError("yume_amd.clip: the images must be on the device — this path has no CPU fallback")
        nblk = self.num_layers - 1 if use_31_block else self.num_layers
        return torch.stack([self.engine.encode(img, nblk) for img in x]).to(x.dtype)


class ClipEngine:
    def __init__(self, model):
        self.model = model
        self._key = None
        self._bufs = {}

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
        self.dev = m.pos_embedding.device
        if self.dev.type != "cuda":
            raise RuntimeError("yume_amd.clip: the model must live on the device — this path has no CPU fallback")
        C, H = m.dim, m.num_heads
        hd = C // H
        if hd > 128 or C % 64:
            raise RuntimeError("yume_amd.clip: head_dim must be <= 128 and dim a multiple of 64")
        f32 = lambda t: t.detach().float().contiguous()
        bf = lambda t: t.detach().to(torch.bfloat16).contiguous()
        kp = m.patch_embedding.weight.shape[1] * m.patch_size ** 2
        wpe = torch.zeros((C, _round_up(kp, 64)), dtype=torch.bfloat16, device=self.dev)
        wpe[:, :kp] = m.patch_embedding.weight.detach().reshape(C, kp).to(torch.bfloat16)
        blocks = []
        for blk in m.transformer:
            a = blk.attn
            # [3, H, hd, C] -> [3, H, 128, C] with zero rows; same for the bias
            w = torch.zeros((3, H, 128, C), dtype=torch.float32, device=self.dev)
            w[:, :, :hd] = a.to_qkv.weight.detach().float().view(3, H, hd, C)
            b = torch.zeros((3, H, 128), dtype=torch.float32, device=self.dev)
            b[:, :, :hd] = a.to_qkv.bias.detach().float().view(3, H, hd)
            wp = torch.zeros((C, H, 128), dtype=torch.float32, device=self.dev)
            wp[:, :, :hd] = a.proj.weight.detach().float().view(C, H, hd)
            blocks.append(dict(n1w=f32(blk.norm1.weight), n1b=f32(blk.norm1.bias), n2w=f32(blk.norm2.weight), n2b=f32(blk.norm2.bias),
                               wqkv=bf(w.view(3 * H * 128, C)), bqkv=b.view(-1).contiguous(), wo=bf(wp.view(C, H * 128)),
                               bo=f32(a.proj.bias), w1=bf(blk.mlp[0].weight), b1=f32(blk.mlp[0].bias), w2=bf(blk.mlp[2].weight),
                               b2=f32(blk.mlp[2].bias), eps=blk.norm_eps))
        self.P = dict(wpe=wpe, kp=kp, cls=f32(m.cls_embedding).view(1, C), pos=f32(m.pos_embedding).view(-1, C),
                      prew=f32(m.pre_norm.weight), preb=f32(m.pre_norm.bias), blocks=blocks)
        self._key = key

    @torch.no_grad()
    def encode(self, img, nblk):
        """img [3, S, S] (normalised) -> fp32 [1 + patches, dim] after `nblk` blocks."""
        self.ensure_packed()
        m, P = self.model, self.P
        C, H, ps = m.dim, m.num_heads, m.patch_size
        g = m.image_size // ps
        L = g * g + 1
        Lp = _round_up(L, 8)
        scale = (C // H) ** -0.5
        # patch embedding: stride = kernel conv == GEMM over the flattened patches (layout change in torch)
        a = self._buf("pe_a", (g * g, P["wpe"].shape[1]), torch.bfloat16, zero=True)
        a[:, :P["kp"]] = img.float().view(3, g, ps, g, ps).permute(1, 3, 0, 2, 4).reshape(g * g, P["kp"]).to(torch.bfloat16)
        xs = self._buf("xs", (L, C), torch.float32)
        ops.gemm_bf16(a, P["wpe"], None, xs[1:], ops.EPI_F32)
        xs[0] = P["cls"][0]
        xs += P["pos"]
        x = self._buf("x", (L, C), torch.float32)
        ops.adaln_modulate(xs, P["prew"], P["preb"], 0, None, False, x, 1, eps=m.norm_eps)          # pre_norm (affine), fp32 out
        h = self._buf("h", (L, C), torch.bfloat16)
        qk = self._buf("qk", (L, 2 * H * 128), torch.bfloat16)
        vt = self._buf("vt", (H * 128, Lp), torch.bfloat16, zero=True)
        att = self._buf("att", (L, H * 128), torch.bfloat16)
        ff = self._buf("ff", (L, P["blocks"][0]["w1"].shape[0]), torch.bfloat16)
        for d in P["blocks"][:nblk]:
            ops.adaln_modulate(x, d["n1w"], d["n1b"], 0, None, False, h, 0, eps=d["eps"])
            ops.gemm_bf16(h, d["wqkv"], d["bqkv"], qk, ops.EPI_BF16_SPLITT, out_t=vt, n_split=2 * H * 128)
            ops.attn_fwd(qk[:, :H * 128], qk[:, H * 128:], vt, att, L, L, H, scale=scale)
            ops.gemm_small_m(att, d["wo"], d["bo"], x, ops.EPI_RESID)
            ops.adaln_modulate(x, d["n2w"], d["n2b"], 0, None, False, h, 0, eps=d["eps"])
            ops.gemm_small_m(h, d["w1"], d["b1"], ff, ops.EPI_BF16_GELU_ERF)
            ops.gemm_small_m(ff, d["w2"], d["b2"], x, ops.EPI_RESID)
        return x.clone()


def clip_vit_h_14_visual(dtype=torch.float32, device="cpu", **kwargs):
    """The vision half of reference clip.py:471-498 clip_xlm_roberta_vit_h_14."""
    cfg = dict(image_size=224, patch_size=14, dim=1280, mlp_ratio=4, out_dim=1024, num_heads=16, num_layers=32, pool_type="token",
               pre_norm=True, post_norm=False, activation="gelu", norm_eps=1e-5)
    cfg.update(kwargs)
    with torch.device(device):
        model = VisionTransformer(**cfg)
    return model.to(dtype=dtype, device=device)


class CLIPModel:
    """reference clip.py:501-542: `.visual(videos)` with videos a list of [3, T, H, W] in [-1, 1] -> [sum T, 257, 1280]."""

    def __init__(self, dtype=torch.float16, device="cuda", checkpoint_path=None, tokenizer_path=None, model=None):
        self.dtype, self.device = dtype, device
        self.checkpoint_path, self.tokenizer_path = checkpoint_path, tokenizer_path
        if model is None:
            model = clip_vit_h_14_visual(dtype=dtype, device=device)
            if checkpoint_path is not None:
                sd = torch.load(checkpoint_path, map_location="cpu")
                model.load_state_dict({k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")})
        self.model = _Visual(model.eval().requires_grad_(False).to(device))

    def visual(self, videos):
        vm = self.model.visual
        size = (vm.image_size,) * 2
        videos = torch.cat([F.interpolate(u.to(self.device).float().transpose(0, 1), size=size, mode="bicubic", align_corners=False)
                            for u in videos])                                      # preprocessing glue stays in torch (clip.py:529-536)
        videos = videos.mul(0.5).add(0.5)
        mean = torch.tensor(_MEAN, device=videos.device).view(1, 3, 1, 1)
        std = torch.tensor(_STD, device=videos.device).view(1, 3, 1, 1)
        return vm((videos - mean) / std, use_31_block=True)


class _Visual(nn.Module):
    """so that `clip.model.visual`, `clip.model.to(...)`, `clip.model.eval()` keep working as in wan/image2video.py:208-209,338"""

    def __init__(self, visual):
        super().__init__()
        self.visual = visual
