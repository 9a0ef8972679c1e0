#!/usr/bin/env python3
"""End-to-end Yume-5B-720P sampling on one MI355X with random-init weights (no checkpoints / tokenizer in this environment):
token ids -> umT5-XXL encoder -> conditioning clip -> Wan2.2 VAE encode -> FramePack chunks of Euler steps on the 5B DiT ->
Wan2.2 VAE decode -> uint8 frames. Everything on the device path of yume_amd (fails loudly without libyume_hip.so).

    python examples/sample_5b_synthetic.py [--chunks 2] [--steps 8] [--small]

It mirrors the order of operations of fastvideo/sample/sample_5b.py (T5: :1197-1210, VAE encode of the conditioning
frames :487/:892, the chunk loop :920-1097, decode + save_video :491-500). `--small` uses test-size models (seconds)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from yume_amd import sampling, synth, t5  # noqa: E402
from yume_amd.video import VideoProcessor  # noqa: E402
from yume_amd.wan23.modules.model import WanModel  # noqa: E402
from yume_amd.wan23.modules.vae2_2 import Wan2_2_VAE, WanVAE_  # noqa: E402


def randomize_(module, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for k, p in module.named_parameters():
            if k.endswith("gamma") or "norm" in k and k.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g, device="cuda"))
            elif k.endswith("bias"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g, device="cuda"))
            elif p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g, device="cuda") * p[0].numel() ** -0.5)
            else:
                p.copy_(0.5 * torch.randn(p.shape, generator=g, device="cuda"))


def tick(msg, t0):
    torch.cuda.synchronize()
    print(f"[{time.perf_counter() - t0:7.2f} s] {msg}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=2)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--small", action="store_true")
    args = ap.parse_args()
    dev = "cuda"
    t0 = time.perf_counter()
    if args.small:
        dcfg = synth.tiny_cfg("wan23")
        tcfg = synth.tiny_t5_cfg(dim=dcfg["text_dim"], heads=4, ffn=512, layers=2)
        vcfg, hw, frames = synth.tiny_vae_cfg("2.2"), (64, 96), 17
    else:
        dcfg, tcfg, vcfg, hw, frames = dict(synth.CFG_5B), dict(synth.T5_CFG_XXL), synth.VAE_CFG_22, (704, 1280), 17
    with torch.device(dev):
        enc = t5.T5Encoder(tcfg["vocab"], tcfg["dim"], tcfg["dim_attn"], tcfg["dim_ffn"], tcfg["num_heads"], tcfg["num_layers"],
                           tcfg["num_buckets"], shared_pos=False).to(torch.bfloat16)
        dit = WanModel(**dcfg)
        vae_m = WanVAE_(dim=vcfg["dim"], dec_dim=vcfg["dec_dim"], z_dim=vcfg["z_dim"], temperal_downsample=vcfg["temperal_downsample"])
    randomize_(enc, 1)
    synth.randomize_module_(dit, seed=2)
    randomize_(vae_m, 3)
    dit = dit.to(torch.bfloat16).eval().requires_grad_(False)
    text = t5.T5EncoderModel(text_len=dcfg["text_len"], device=dev, model=enc)
    vae = Wan2_2_VAE(z_dim=vcfg["z_dim"], device=dev, model=vae_m)
    tick("models built (random weights)", t0)

    g = torch.Generator(device=dev).manual_seed(0)
    L = dcfg["text_len"]
    ids = torch.randint(1, tcfg["vocab"], (args.chunks, L), device=dev, generator=g)
    mask = torch.zeros(args.chunks, L, dtype=torch.long, device=dev)
    for b in range(args.chunks):
        mask[b, :min(L, 20 + 11 * b)] = 1                                   # prompts of different lengths
    contexts = [c.float() for c in text.encode_ids(ids, mask)]
    tick(f"umT5 encoded {args.chunks} prompts: {[tuple(c.shape) for c in contexts]}", t0)

    clip = torch.rand(3, frames, *hw, device=dev, generator=g) * 2 - 1     # the conditioning clip
    hist = vae.encode([clip])[0]
    tick(f"VAE encoded the {frames}-frame conditioning clip -> latents {tuple(hist.shape)}", t0)

    def on_chunk(k, latent):
        tick(f"chunk {k}: {args.steps} denoise steps done, history now {latent.shape[1]} latent frames", t0)
    all_lat, videos = sampling.long_video_5b(dit, vae, hist, contexts, steps=args.steps, generator=g, on_chunk=on_chunk)
    tick(f"decoded {len(videos)} chunks: {[tuple(v.shape) for v in videos]}", t0)

    u8 = VideoProcessor(vae_scale_factor=8).postprocess_video(torch.cat(videos, dim=1).unsqueeze(0), output_type="uint8")
    host = u8.cpu()
    tick(f"uint8 frames on the host: {tuple(host.shape)}, mean {host.float().mean():.1f}", t0)
    assert torch.isfinite(all_lat).all()


if __name__ == "__main__":
    main()
