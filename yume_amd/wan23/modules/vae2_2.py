"""Drop-in Wan2.2 causal 3D VAE (z=48, stride 4x16x16), MI355X-native.

Keeps the reference call surface of wan23/modules/vae2_2.py: `WanVAE_(dim, dec_dim, z_dim, dim_mult, num_res_blocks,
attn_scales, temperal_downsample, dropout)` with `.encode(x, scale)`, `.decode(z, scale)`, `.clear_cache()` and the same
state_dict keys (so the `.pth` checkpoint loads with `load_state_dict(..., assign=True)`, :897-904), and the
`Wan2_2_VAE` wrapper with `encode(list) -> list` / `decode(list) -> list` (:1045-1072). The module tree only owns the
parameters; all arithmetic runs in the HIP kernels through yume_amd.vae.VaeEngine (no PyTorch fallback).
"""
import logging

import torch
import torch.nn as nn

from ... import synth
from ...vae import VaeEngine, build_param_tree

__all__ = ["Wan2_2_VAE", "WanVAE_"]

_MEAN = [-0.2289, -0.0052, -0.1323, -0.2339, -0.2799, 0.0174, 0.1838, 0.1557, -0.1382, 0.0542, 0.2813, 0.0891, 0.1570,
         -0.0098, 0.0375, -0.1825, -0.2246, -0.1207, -0.0698, 0.5109, 0.2665, -0.2108, -0.2158, 0.2502, -0.2055, -0.0322,
         0.1109, 0.1567, -0.0729, 0.0899, -0.2799, -0.1230, -0.0313, -0.1649, 0.0117, 0.0723, -0.2839, -0.2083, -0.0520,
         0.3748, 0.0152, 0.1957, 0.1433, -0.2944, 0.3573, -0.0548, -0.1681, -0.0667]
_STD = [0.4765, 1.0364, 0.4514, 1.1677, 0.5313, 0.4990, 0.4818, 0.5013, 0.8158, 1.0344, 0.5894, 1.0901, 0.6885, 0.6165,
        0.8454, 0.4978, 0.5759, 0.3523, 0.7135, 0.6804, 0.5833, 1.4146, 0.8986, 0.5659, 0.7069, 0.5338, 0.4889, 0.4917,
        0.4069, 0.4999, 0.6866, 0.4093, 0.5709, 0.6065, 0.6415, 0.4944, 0.5726, 1.2042, 0.5458, 1.6887, 0.3971, 1.0600,
        0.3943, 0.5537, 0.5444, 0.4089, 0.7468, 0.7744]


class _VaeBase(nn.Module):
    """shared by the 2.2 and 2.1 drop-ins: parameter tree + engine + the reference's encode/decode(x, scale) surface."""
    _vae_version = "2.2"

    def _setup(self, cfg):
        self.cfg = cfg
        tree = build_param_tree(synth.vae_param_shapes(cfg))
        for name, child in tree.named_children():
            self.add_module(name, child)
        self.z_dim = cfg["z_dim"]
        self._engine = None

    @property
    def engine(self):
        if self._engine is None:
            self._engine = VaeEngine(self, self.cfg)
        return self._engine

    def clear_cache(self):
        """the per-call chunk caches live inside encode()/decode(); nothing persists between calls."""

    def encode(self, x, scale):
        """x [B, 3, T, H, W] -> mu [B, z, T', h, w] scaled as (mu - scale[0]) * scale[1] (vae2_2.py:797-829)."""
        outs = [self.engine.encode(u, sub=scale[0], mul=scale[1]) for u in x]
        return torch.stack(outs)

    def decode(self, z, scale):
        """z [B, z, T, h, w] -> x [B, 3, 1+4(T-1), H, W] from z / scale[1] + scale[0] (vae2_2.py:831-860). Not clamped."""
        s1 = scale[1]
        mul = (1.0 / s1) if isinstance(s1, torch.Tensor) else 1.0 / float(s1)
        outs = [self.engine.decode(u, mul=mul, add=scale[0], clamp=False) for u in z]
        return torch.stack(outs)

    def forward(self, x, scale=(0, 1)):
        mu = self.encode(x, scale)
        return self.decode(mu, scale), mu


class WanVAE_(_VaeBase):
    def __init__(self, dim=160, dec_dim=256, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_downsample=[True, True, False], dropout=0.0):
        super().__init__()
        if attn_scales:
            raise NotImplementedError("attn_scales is empty in every Yume/Wan configuration")
        self._setup(dict(version="2.2", dim=dim, dec_dim=dec_dim, z_dim=z_dim, dim_mult=list(dim_mult),
                         num_res_blocks=num_res_blocks, temperal_downsample=list(temperal_downsample), patch=2, in_ch=12))


def _video_vae(pretrained_path=None, z_dim=16, dim=160, device="cpu", **kwargs):
    cfg = dict(dim=dim, z_dim=z_dim, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
               temperal_downsample=[True, True, True], dropout=0.0)
    cfg.update(**kwargs)
    with torch.device("meta"):
        model = WanVAE_(**cfg)
    logging.info(f"loading {pretrained_path}")
    model.load_state_dict(torch.load(pretrained_path, map_location=device), assign=True)
    return model


class Wan2_2_VAE:
    def __init__(self, z_dim=48, c_dim=160, vae_pth=None, dim_mult=[1, 2, 4, 4], temperal_downsample=[False, True, True],
                 dtype=torch.float, device="cuda", model=None):
        self.dtype, self.device = dtype, device
        mean = torch.tensor(_MEAN, dtype=dtype, device=device)
        std = torch.tensor(_STD, dtype=dtype, device=device)
        self.scale = [mean, 1.0 / std]
        if model is None:   # `model=` lets tests/bench inject a random-init WanVAE_ (no checkpoint offline)
            model = _video_vae(pretrained_path=vae_pth, z_dim=z_dim, dim=c_dim, dim_mult=dim_mult,
                               temperal_downsample=temperal_downsample)
        self.model = model.eval().requires_grad_(False).to(device)

    def encode(self, videos, cache=True):
        if not isinstance(videos, list):
            logging.info("videos should be a list")
            return None
        return [self.model.encode(u.unsqueeze(0), self.scale).float().squeeze(0) for u in videos]

    def decode(self, zs):
        if not isinstance(zs, list):
            logging.info("zs should be a list")
            return None
        # model.decode(z, scale) followed by .clamp_(-1, 1): the clamp is fused into the output layout kernel
        return [self.model.engine.decode(u, mul=1.0 / self.scale[1], add=self.scale[0], clamp=True) for u in zs]
