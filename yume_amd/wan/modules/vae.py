"""Drop-in Wan2.1 causal 3D VAE (z=16, stride 4x8x8), MI355X-native — reference wan/modules/vae.py
(identical copies: wan23/modules/vae2_1.py, fastvideo/models/hunyuan/modules/vae.py).

`WanVAE_(dim, z_dim, dim_mult, num_res_blocks, attn_scales, temperal_downsample, dropout)` with
`.encode(x, scale)` / `.decode(z, scale)` / `.clear_cache()` and the `WanVAE` wrapper (`encode(list)`, `decode(list)`,
vae.py:619-663). Parameters only; arithmetic in the HIP kernels via yume_amd.vae.VaeEngine.
"""
import logging

import torch

from ...wan23.modules.vae2_2 import _VaeBase

__all__ = ["WanVAE", "WanVAE_"]

_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632, -0.1922,
         -0.9497, 0.2503, -0.2921]
_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253,
        2.8251, 1.9160]


class WanVAE_(_VaeBase):
    def __init__(self, dim=128, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_downsample=[True, True, False], dropout=0.0):
        super().__init__()
        if attn_scales:
            raise NotImplementedError("attn_scales is empty in every Yume/Wan configuration")
        self._setup(dict(version="2.1", dim=dim, dec_dim=dim, z_dim=z_dim, dim_mult=list(dim_mult),
                         num_res_blocks=num_res_blocks, temperal_downsample=list(temperal_downsample), patch=1, in_ch=3))


def _video_vae(pretrained_path=None, z_dim=None, device="cpu", **kwargs):
    cfg = dict(dim=96, z_dim=z_dim, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
               temperal_downsample=[False, True, True], dropout=0.0)
    cfg.update(**kwargs)
    with torch.device("meta"):
        model = WanVAE_(**cfg)
    logging.info(f"loading {pretrained_path}")
    model.load_state_dict(torch.load(pretrained_path, map_location=device), assign=True)
    return model


class WanVAE:
    def __init__(self, z_dim=16, vae_pth="cache/vae_step_411000.pth", dtype=torch.float, device="cuda", model=None):
        self.dtype, self.device = dtype, device
        self.mean = torch.tensor(_MEAN, dtype=dtype, device=device)
        self.std = torch.tensor(_STD, dtype=dtype, device=device)
        self.scale = [self.mean, 1.0 / self.std]
        if model is None:
            model = _video_vae(pretrained_path=vae_pth, z_dim=z_dim)
        self.model = model.eval().requires_grad_(False).to(device)

    def encode(self, videos):
        return [self.model.encode(u.unsqueeze(0), self.scale).float().squeeze(0) for u in videos]

    def decode(self, zs):
        return [self.model.engine.decode(u, mul=1.0 / self.scale[1], add=self.scale[0], clamp=True) for u in zs]
