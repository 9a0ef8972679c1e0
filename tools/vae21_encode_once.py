#!/usr/bin/env python3
"""Workload of a rocprofv3 --kernel-trace --stats pass: two Wan2.1 49-frame 544x960 encodes (random weights)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import synth
from yume_amd.wan.modules.vae import WanVAE, WanVAE_
dev = "cuda"
cfg = synth.VAE_CFG_21
with torch.device(dev):
    m = WanVAE_(dim=cfg["dim"], z_dim=cfg["z_dim"], temperal_downsample=cfg["temperal_downsample"])
m.load_state_dict(synth.make_vae_state_dict(cfg, seed=5, device=dev), strict=True)
vae = WanVAE(device=dev, model=m)
video = torch.rand(3, 49, 544, 960, device=dev) * 2 - 1
for _ in range(2):
    out = vae.encode([video])[0]
torch.cuda.synchronize()
print(tuple(out.shape), bool(torch.isfinite(out).all()))
