#!/usr/bin/env python3
"""Full-size Wan2.2 VAE decode/encode timing on one MI355X (random weights). Writes gpurun_out/vae_probe.json."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import synth
from yume_amd.wan23.modules.vae2_2 import Wan2_2_VAE, WanVAE_

def main():
    dev = "cuda"
    cfg = synth.VAE_CFG_22
    with torch.device(dev):
        m = WanVAE_(dim=cfg["dim"], dec_dim=cfg["dec_dim"], z_dim=cfg["z_dim"], temperal_downsample=cfg["temperal_downsample"])
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith("gamma"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g, device=dev))
            elif k.endswith("bias"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g, device=dev))
            else:
                fan_in = p[0].numel()
                p.copy_((torch.rand(p.shape, generator=g, device=dev) * 2 - 1) * (3.0 / fan_in) ** 0.5)
    vae = Wan2_2_VAE(device=dev, model=m)
    res = {}
    nlat = int(os.environ.get("YUME_VAE_LATENTS", "8"))
    z = torch.randn(48, nlat, 44, 80, device=dev, generator=g)
    # the reference's walk (one latent frame per decoder pass) next to the grouped passes: same outputs?
    eng = m.engine
    group = eng.group
    eng.group = 1
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ref = vae.decode([z])[0]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"decode {nlat} latents, one per pass -> {tuple(ref.shape)}: {dt*1e3:.1f} ms  ({nlat/dt:.2f} latents/s)", flush=True)
    res["decode_ms_group1"] = dt * 1e3
    eng.group = group
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = vae.decode([z])[0]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"decode {nlat} latents, group {group} -> {tuple(out.shape)}: {dt*1e3:.1f} ms  ({nlat/dt:.2f} latents/s)  finite={bool(torch.isfinite(out).all())}", flush=True)
    d = (out - ref).abs().max().item()
    print(f"grouped vs one-per-pass: max abs diff {d:.3e} (output rms {ref.pow(2).mean().sqrt().item():.3f}), equal bits: {bool(torch.equal(out, ref))}", flush=True)
    res["group_vs_walk_maxabs"] = d
    del ref
    res["decode_ms"] = dt * 1e3; res["decode_latents_per_s"] = nlat / dt
    res["decode_tflops"] = (485.04 if nlat == 8 else None) and 485.04 / dt / 1e0 / 1e0 if nlat == 8 else None
    video = torch.rand(3, 17, 704, 1280, device=dev, generator=g) * 2 - 1
    eng.group = 1
    lat1 = vae.encode([video])[0]
    eng.group = group
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lat = vae.encode([video])[0]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"encode 17 frames -> {tuple(lat.shape)}: {dt*1e3:.1f} ms", flush=True)
    res["encode17_ms"] = dt * 1e3
    print(f"encode grouped vs chunk walk: max abs diff {(lat - lat1).abs().max().item():.3e}, equal bits: {bool(torch.equal(lat, lat1))}", flush=True)
    print("peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "vae_probe.json"), "w"))

if __name__ == "__main__":
    main()
