#!/usr/bin/env python3
"""bench.py — denoise-steps/sec of the Yume-5B-720P ODE sampler hot loop on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one Euler step of fastvideo/sample/sample_5b.py:960-1034 on BASELINE.json configs[1]: one
WanModel.forward (no CFG) of the random-init Yume-5B-720P model on a 33-frame 704x1280 clip (latent
[48,13,44,80], FramePack latent_frame_zero=8, L = 9460 tokens, 512 padded text tokens, per-token timesteps)
plus the Euler update of the 8 new latent frames. Inputs are resident in HBM before the timed region.
Multi-GPU: every rank runs its own independent chain (its own prompt/noise), exactly the reference's
`index = (step-1)*world_size + rank` sharding — no collective inside the loop ("scaling": "weak");
value = total steps of all ranks / max-over-ranks time.

The JSON line also carries
  roofline     : the dominant kernel (ffn.0 bf16 MFMA GEMM, 9460x14336x3072) timed with HIP events on the launch
                 stream inside the timed steps, against the 2.5 PFLOP/s dense bf16 MFMA peak.
  cpu_baseline : the CPU oracle restatement of the reference (oracle/dit.py) timed on the host cores on a bounded
                 sample (one DiT block at the full L, extrapolated to 30 blocks), rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0     # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16


def flops_fwd_5b(L, C=3072, ffn=14336, n=30, Lc=512, cin=48, cout=48):
    """SURVEY.md §8(d) F_fwd(L) (2 flop per MAC)."""
    blk = 8 * L * C * C + 4 * L * L * C + 4 * L * C * C + 4 * Lc * C * C + 4 * L * Lc * C + 4 * L * C * ffn
    return n * blk + 2 * L * (cin * 4) * C + 2 * 512 * (4096 * C + C * C) + 2 * L * (256 * C + C * C + 6 * C * C) + 2 * L * C * 4 * cout


def pmc_traffic_bytes():
    """HBM/fabric bytes per launch of the dominant kernel from the committed rocprofv3 PMC pass
    (profiles/r1_pmc_dominant_kernels.csv: FETCH_SIZE and WRITE_SIZE in KiB; FETCH_SIZE doubled per the gfx950
    calibration in MI355X_MICROARCH.md, confirmed on the adaLN kernel: 2*FETCH = 4*L*C exactly)."""
    import csv
    path = os.path.join(ROOT, "profiles", "r1_pmc_dominant_kernels.csv")
    try:
        for r in csv.DictReader(open(path)):
            if "gemm256_kernel<1" in r["kernel"]:
                return (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0
    except Exception:  # noqa: BLE001
        pass
    return None


def vae_decode_rate(dev, z8):
    """Wan2.2 VAE decode of one chunk (8 latents [48,8,44,80] -> 29 frames 704x1280), random-init weights:
    the second half of BASELINE.json's metric ("VAE dec latents/s"). Not part of `value`."""
    from yume_amd import synth
    from yume_amd.wan23.modules.vae2_2 import Wan2_2_VAE, WanVAE_
    cfg = synth.VAE_CFG_22
    with torch.device(dev):
        m = WanVAE_(dim=cfg["dim"], dec_dim=cfg["dec_dim"], z_dim=cfg["z_dim"], temperal_downsample=cfg["temperal_downsample"])
    g = torch.Generator(device=dev).manual_seed(5)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith("gamma"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g, device=dev))
            elif k.endswith("bias"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g, device=dev))
            else:
                p.copy_((torch.rand(p.shape, generator=g, device=dev) * 2 - 1) * (3.0 / p[0].numel()) ** 0.5)
    vae = Wan2_2_VAE(device=dev, model=m)
    vae.decode([z8])                                   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = vae.decode([z8])[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    return {"latents_per_s": z8.shape[1] / dt, "ms_per_chunk": dt * 1e3, "chunk": "8 latents 48x8x44x80 -> 29 frames 704x1280",
            "tflop_per_chunk": 485.04, "tflops": 485.04 / dt}


def cpu_baseline(cfg, L, n_hist, seconds_budget=30.0):
    """Reference restatement on the host cores: one full-width block at the full sequence length."""
    from oracle import dit as odit
    from yume_amd import synth
    torch.set_num_threads(os.cpu_count() or 1)
    c1 = dict(cfg)
    c1["num_layers"] = 1
    sd = {k: v for k, v in synth.make_dit_state_dict(c1, "wan23", seed=0, pyramid=()).items() if k.startswith("blocks.0.")}
    C = cfg["dim"]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(L, C, generator=g)
    e6 = torch.randn(L, 6, C, generator=g) * 0.1
    ctx = torch.randn(512, C, generator=g)
    tabs = odit.rope_axes(128)
    rope = odit.rope_grid(tabs, 1, 1, L, 0) if L <= 1024 else torch.polar(torch.ones(L, 64, dtype=torch.float64),
                                                                           torch.randn(L, 64, generator=g).double())
    # fp32 attention for the timing leg (the fp64 exact-softmax of the parity oracle would dominate the CPU time)
    orig = odit.attention

    def attn32(q, k, v):
        return torch.nn.functional.scaled_dot_product_attention(q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1)).transpose(0, 1)
    odit.attention = attn32
    try:
        t0 = time.time()
        with torch.no_grad():
            odit.block_forward(sd, "blocks.0.", x, e6, rope, ctx, c1, "wan23")
        dt = time.time() - t0
    finally:
        odit.attention = orig
    step_s = dt * cfg["num_layers"]
    return {"value": 1.0 / step_s, "unit": "denoise-steps/sec", "cores": os.cpu_count(), "kind": "port",
            "sample": f"1 of {cfg['num_layers']} DiT blocks at L={L} (fp32, {dt:.2f} s), extrapolated x{cfg['num_layers']}; embed/head excluded"}


def bench_14b(args, rank, world, dev):
    """BASELINE configs[2]: Yume-I2V-14B-540P random-init, 65-frame 544x960 clip (latent [16,17,68,120] + y[20,...]),
    FramePack (rand_num_img=0.6, latent_frame_zero=9), CFG 5.0 -> two forwards per step, history re-noised each step
    (fastvideo/sample/sample.py:745-790). L = 27810 tokens, 2 x 1319.3 TFLOP per step."""
    from yume_amd import distributed as ydist
    from yume_amd import framepack, sampling, synth
    from yume_amd.wan.modules.model import WanModel
    cfg = dict(synth.CFG_14B)
    if args.layers:
        cfg["num_layers"] = args.layers
    F, H, W, lfz, S, shift = 17, 68, 120, 9, 50, 3.0
    with torch.device(dev):
        model = WanModel(**cfg).attach_pyramid()
    if rank == 0:
        synth.randomize_module_(model, seed=0)
    model = model.to(torch.bfloat16).eval().requires_grad_(False)
    ydist.broadcast_module_(model, src=0)
    L = framepack.pack_plan(F, H, W, lfz, F - 9).seq_len
    g = torch.Generator(device=dev).manual_seed(2000 + rank)
    clean = torch.randn((16, F, H, W), generator=g, device=dev)
    noise = torch.randn((16, F, H, W), generator=g, device=dev)
    y = [torch.randn((20, F, H, W), generator=g, device=dev)]
    clip = torch.randn((1, 257, 1280), generator=g, device=dev)
    arg_c = dict(context=[torch.randn((77, 4096), generator=g, device=dev)], clip_fea=clip, seq_len=L, y=y)
    arg_null = dict(context=[torch.randn((77, 4096), generator=g, device=dev)], clip_fea=clip, seq_len=L, y=y)
    sig = synth.sampling_sigmas(S, shift)
    vel = sampling.make_velocity_14b(model, arg_c, arg_null, sig, guide=5.0, rand_num_img=0.6, lfz=lfz)
    hist = sampling.renoised_history(clean[:, :-lfz], noise[:, :-lfz], sig)
    latent = noise.clone()

    def step(i, latent):
        k = i % S
        v = vel(latent, k)
        nxt = sig[k + 1] if k + 1 < S else 0.0
        x = latent[:, -lfz:] + (nxt - sig[k]) * v[:, -lfz:]
        return torch.cat([hist(min(S - 1, k + 1)), x], dim=1)

    for i in range(args.warmup):
        latent = step(i, latent)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        latent = step(i, latent)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(latent).all()
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = float(tmax.item())
    if rank == 0:
        ms = tmax / args.steps * 1e3
        tf = 2 * 1319.3 * (cfg["num_layers"] / 40.0)
        print(json.dumps({"metric": "denoise-steps/sec (Yume-I2V-14B 540P, 65-frame latent, CFG)", "value": world * args.steps / tmax,
                          "unit": "denoise-steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": "Yume-I2V-14B-540P random-init, latent 16x17x68x120 + y, FramePack lfz=9, L=27810, CFG 5.0 "
                                                 "(2 forwards/step), 50-step shift-3 schedule", "num_layers": cfg["num_layers"], "tokens": L},
                          "model_tflop_per_step": tf, "model_tflops_per_gpu": tf / (ms * 1e-3),
                          "peak_mem_gb": torch.cuda.max_memory_allocated() / 1e9}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=0, help="debug only: override num_layers (result is then NOT the named config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the (untimed-in-value) VAE decode measurement")
    ap.add_argument("--sp", action="store_true",
                    help="5b only, not the headline: ONE chain split over the N ranks (Ulysses sequence parallelism, "
                         "SURVEY 8(f).2); value is that chain's steps/s, scaling 'strong'")
    ap.add_argument("--workload", default="5b", choices=["5b", "14b"],
                    help="5b = BASELINE configs[1] (the headline metric, default); 14b = configs[2] (Yume-I2V-14B-540P, 65-frame clip, CFG)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from yume_amd import framepack, synth
    from yume_amd.wan23.modules.model import WanModel

    if args.workload == "14b":
        return bench_14b(args, rank, world, dev)
    cfg = dict(synth.CFG_5B)
    if args.layers:
        cfg["num_layers"] = args.layers
    F, H, W, lfz, steps_total, shift = 13, 44, 80, 8, 50, 7.0
    from yume_amd import distributed as ydist
    with torch.device(dev):
        model = WanModel(**cfg)
    if rank == 0:
        synth.randomize_module_(model, seed=0)
    model = model.to(torch.bfloat16).eval().requires_grad_(False)   # sample_5b.py:1241 casts the transformer to bf16
    n_bcast = ydist.broadcast_module_(model, src=0)   # replicated weights: one-time RCCL broadcast, flat 1 GiB buckets
    plan = framepack.pack_plan(F, H, W, lfz)
    L = plan.seq_len

    sp = bool(args.sp) and world > 1
    if sp:
        model.enable_sequence_parallel()                                 # all ranks: the same chain
    g = torch.Generator(device=dev).manual_seed(1000 + (0 if sp else rank))   # each rank: its own prompt / noise
    hist = torch.randn((48, F - lfz, H, W), generator=g, device=dev)
    latent = torch.cat([hist, torch.randn((48, lfz, H, W), generator=g, device=dev)], dim=1)
    context = [torch.randn((77, 4096), generator=g, device=dev)]
    sig = synth.sampling_sigmas(steps_total, shift)
    zeros_hist = torch.zeros(plan.n_hist_tok, dtype=torch.float64, device=dev)
    ones_new = torch.ones(plan.n_new_tok, dtype=torch.float64, device=dev)

    def step(i, latent):
        s = sig[i % steps_total]
        s_next = sig[i % steps_total + 1] if (i % steps_total) + 1 < steps_total else 0.0
        t = torch.cat([zeros_hist, ones_new * (s * 1000.0)]).unsqueeze(0)              # sample_5b.py:965-972
        pred = model([latent], t=t, context=context, seq_len=L, latent_frame_zero=lfz, flag=True)[0]
        new = latent[:, -lfz:] + (s_next - s) * pred                                    # :987-990
        return torch.cat([hist, new], dim=1)                                            # :1031-1034

    for i in range(args.warmup):
        latent = step(i, latent)
    model.engine.prof = []
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        latent = step(i, latent)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, model.engine.prof = model.engine.prof, None
    assert torch.isfinite(latent).all(), "non-finite latents"
    # results of every chain are gathered at chunk end (the only other collective of the run)
    checks = ydist.gather_scalars(float(latent[:, -lfz:].double().abs().mean()), device=dev)

    # SURVEY §8(f).1 (step-invariant conditioning cached across steps): reported beside the headline, never as `value`
    cached_ms = None
    if rank == 0:
        model.engine.cache_context = True
        lat2 = step(0, latent)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for i in range(1, 1 + max(2, min(args.steps, 5))):
            lat2 = step(i, lat2)
        torch.cuda.synchronize()
        cached_ms = (time.perf_counter() - tc) / max(2, min(args.steps, 5)) * 1e3
        model.engine.cache_context = False
        del lat2

    vae_res = None
    if not args.no_vae and rank == 0:
        vae_res = vae_decode_rate(dev, latent[:, -lfz:].float())

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = float(tmax.item())

    if rank == 0:
        gemm_ms = sum(a.elapsed_time(b) for a, b in prof) / max(1, len(prof))
        gemm_flop = 2.0 * L * cfg["ffn_dim"] * cfg["dim"]
        achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        ms_per_step = tmax / args.steps * 1e3
        out = {
            "metric": "denoise-steps/sec (Yume-5B 720P, 33-frame latent)",
            "value": (1 if sp else world) * args.steps / tmax, "unit": "denoise-steps/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if sp else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Yume-5B-720P random-init, 33-frame 704x1280 clip (latent 48x13x44x80, FramePack "
                                   "lfz=8, L=9460), ODE Euler steps of a 50-step shift-7 schedule, no CFG, one chain per GPU",
                       "num_layers": cfg["num_layers"], "tokens": L, "parallelism": (f"sp{world} (one chain, Ulysses all-to-all, replicated weights)" if sp
                                       else f"dp{world} (independent chains, replicated weights)")},
            "chain_checksums": checks, "weight_broadcast_collectives": n_bcast,
            "vae_decode": vae_res,
            "cached_context_ms_per_step": cached_ms,
            "model_tflop_per_step": flops_fwd_5b(L, n=cfg["num_layers"]) / 1e12,
            "model_tflops_per_gpu": flops_fwd_5b(L, n=cfg["num_layers"]) / 1e12 / (ms_per_step * 1e-3),
            "roofline": {"bound": "mfma", "kernel": "ffn.0 GEMM 9460x14336x3072 + bias + GELU: gemm256_kernel<EPI_BF16_GELU, PlainA> on rows 0..9215 "
                                                       "+ gemm128_kernel on the last 244 rows (one yume_gemm_bf16 call, timed as one launch)",
                         "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": pmc_traffic_bytes(),
                         "launch_ms": gemm_ms, "launches_timed": len(prof)},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, L, plan.n_hist_tok)
            except Exception as e:  # noqa: BLE001 — a baseline failure must not hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "denoise-steps/sec", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
