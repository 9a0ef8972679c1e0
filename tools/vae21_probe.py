#!/usr/bin/env python3
"""Full-size Wan2.1 VAE (14B pipeline) decode/encode timing on one MI355X (random weights)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import synth
from yume_amd.wan.modules.vae import WanVAE, WanVAE_
dev = "cuda"
cfg = synth.VAE_CFG_21
with torch.device(dev):
    m = WanVAE_(dim=cfg["dim"], z_dim=cfg["z_dim"], temperal_downsample=cfg["temperal_downsample"])
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for k, p in m.named_parameters():
        if k.endswith("gamma"): p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g, device=dev))
        elif k.endswith("bias"): p.copy_(0.02 * torch.randn(p.shape, generator=g, device=dev))
        else: p.copy_((torch.rand(p.shape, generator=g, device=dev) * 2 - 1) * (3.0 / p[0].numel()) ** 0.5)
vae = WanVAE(device=dev, model=m)
z = torch.randn(16, 13, 68, 120, device=dev, generator=g)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = vae.decode([z])[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"2.1 decode 13 latents -> {tuple(out.shape)}: {dt*1e3:.1f} ms ({13/dt:.1f} latents/s, {218.6/dt:.0f} TF) finite={bool(torch.isfinite(out).all())}", flush=True)
video = torch.rand(3, 49, 544, 960, device=dev, generator=g) * 2 - 1
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lat = vae.encode([video])[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"2.1 encode 49 frames -> {tuple(lat.shape)}: {dt*1e3:.1f} ms ({130.2/dt:.0f} TF)", flush=True)
print("peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
