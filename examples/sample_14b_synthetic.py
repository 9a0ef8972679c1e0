#!/usr/bin/env python3
"""End-to-end Yume-I2V-14B-540P chunk on one MI355X with random-init weights: conditioning clip -> CLIP ViT-H/14 vision tower
(clip_fea) + Wan2.1 VAE encode (history latents, y) + umT5-XXL (prompt and negative prompt) -> FramePack chunk of CFG Euler steps
on the 14B DiT (history re-noised every step) -> Wan2.1 VAE decode -> uint8 frames.

    python examples/sample_14b_synthetic.py [--steps 4] [--small]

Order of operations as in wan/image2video.py:338-402 (conditioning) and fastvideo/sample/sample.py:745-790 (the step loop)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from examples.sample_5b_synthetic import randomize_, tick  # noqa: E402
from yume_amd import clip, framepack, sampling, synth, t5  # noqa: E402
from yume_amd.video import VideoProcessor  # noqa: E402
from yume_amd.wan.modules.model import WanModel  # noqa: E402
from yume_amd.wan.modules.vae import WanVAE, WanVAE_  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--small", action="store_true")
    args = ap.parse_args()
    dev, lfz = "cuda", 9
    t0 = time.perf_counter()
    if args.small:
        dcfg = synth.tiny_cfg("wan")
        tcfg = synth.tiny_t5_cfg(dim=dcfg["text_dim"], heads=4, ffn=512, layers=2)
        ccfg = synth.tiny_clip_cfg(dim=1280 // 4, heads=4, layers=3, image=56)
        ccfg["dim"] = 1280                       # the DiT's img_emb takes 1280-wide CLIP features
        ccfg["num_heads"] = 16
        vcfg, hw, frames = synth.tiny_vae_cfg("2.1"), (64, 96), 65
    else:
        dcfg, tcfg, ccfg = dict(synth.CFG_14B), dict(synth.T5_CFG_XXL), dict(synth.CLIP_CFG_VIT_H)
        vcfg, hw, frames = synth.VAE_CFG_21, (544, 960), 65
    with torch.device(dev):
        enc = t5.T5Encoder(tcfg["vocab"], tcfg["dim"], tcfg["dim_attn"], tcfg["dim_ffn"], tcfg["num_heads"], tcfg["num_layers"],
                           tcfg["num_buckets"], shared_pos=False).to(torch.bfloat16)
        vis = clip.VisionTransformer(**ccfg)
        dit = WanModel(**dcfg).attach_pyramid()
        vae_m = WanVAE_(dim=vcfg["dim"], z_dim=vcfg["z_dim"], temperal_downsample=vcfg["temperal_downsample"])
    randomize_(enc, 1)
    randomize_(vis, 4)
    synth.randomize_module_(dit, seed=2)
    randomize_(vae_m, 3)
    dit = dit.to(torch.bfloat16).eval().requires_grad_(False)
    text = t5.T5EncoderModel(text_len=dcfg["text_len"], device=dev, model=enc)
    clipm = clip.CLIPModel(device=dev, model=vis.eval())
    vae = WanVAE(z_dim=vcfg["z_dim"], device=dev, model=vae_m)
    tick("models built (random weights)", t0)

    g = torch.Generator(device=dev).manual_seed(0)
    L = dcfg["text_len"]
    ids = torch.randint(1, tcfg["vocab"], (2, L), device=dev, generator=g)
    mask = torch.zeros(2, L, dtype=torch.long, device=dev)
    mask[0, :40] = 1
    mask[1, :12] = 1                                                         # prompt, negative prompt
    ctx, ctx_null = [c.float() for c in text.encode_ids(ids, mask)]
    video = torch.rand(3, frames, *hw, device=dev, generator=g) * 2 - 1     # the conditioning clip
    clip_fea = clipm.visual([video[:, -1:].contiguous()])                   # last frame, image2video.py:344
    clean = vae.encode([video])[0]                                          # [16, 17, h, w]
    tick(f"conditioning: umT5 {tuple(ctx.shape)} / {tuple(ctx_null.shape)}, CLIP {tuple(clip_fea.shape)}, VAE latents {tuple(clean.shape)}", t0)

    C, F, H, W = clean.shape
    msk = torch.ones(4, F, H, W, device=dev)
    msk[:, -lfz:] = 0                                                        # frames to generate
    y = [torch.cat([msk, clean * msk[:1]], dim=0)]                           # 4 mask + 16 latent channels (image2video.py:372-386)
    seq_len = framepack.pack_plan(F, H, W, lfz, F - 9).seq_len
    arg_c = dict(context=[ctx], clip_fea=clip_fea, seq_len=seq_len, y=y)
    arg_null = dict(context=[ctx_null], clip_fea=clip_fea, seq_len=seq_len, y=y)
    sig = synth.sampling_sigmas(args.steps, 3.0)
    noise = torch.randn(clean.shape, generator=g, device=dev)
    vel = sampling.make_velocity_14b(dit, arg_c, arg_null, sig, guide=5.0, rand_num_img=0.6, lfz=lfz)
    hist = sampling.renoised_history(clean[:, :-lfz], noise[:, :-lfz], sig)
    latent = sampling.ode_chunk(vel, noise.clone(), sig, lfz, hist)
    tick(f"{args.steps} CFG denoise steps on {seq_len} tokens done", t0)

    frames_out = vae.decode([torch.cat([clean[:, :-lfz], latent[:, -lfz:]], dim=1)])[0]
    u8 = VideoProcessor(vae_scale_factor=8).postprocess_video(frames_out.unsqueeze(0), output_type="uint8").cpu()
    tick(f"decoded and converted: uint8 frames {tuple(u8.shape)}, mean {u8.float().mean():.1f}", t0)
    assert torch.isfinite(latent).all()


if __name__ == "__main__":
    main()
