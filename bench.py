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
  roofline     : the kernel with the LARGEST share of the timed steps (every kernel group of the block is bracketed with HIP
                 events on the launch stream inside the timed region; today that is the self-attention kernel), against the
                 2.5 PFLOP/s dense bf16 MFMA peak; roofline_all lists the other groups the same way.
  cpu_baseline : the CPU oracle restatement of the reference (oracle/dit.py) timed on the host cores on a bounded
                 sample (one DiT block at the full L, extrapolated to 30 blocks), rank 0 / N=1 only.
  parity       : that same CPU block output against the device engine on identical inputs (rel-L2 / max-abs).
  vae_decode   : Wan2.2 decode latents/s on the GPU and the oracle VAE on the host cores (reduced size, FLOP-scaled); `passes`: the other
                 VAE passes of the two pipelines (Wan2.2 17-frame encode, Wan2.1 decode / encode) with their own roofline fractions.
  calibration  : what THIS box sustains, measured right behind the timed steps (yume_amd/calibrate.py: pure-MFMA microkernel on random
                 operands + one fixed 8192^3 launch of the product GEMM); every MFMA-bound group carries frac_of_sustained next to frac.

--workload tts / longvideo / 14b run BASELINE.json configs[3] / [4] / [2] (see their functions); the default is configs[1].
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


PMC_GROUP_FILE = "r6_pmc_gemm_block_shapes.csv"   # (r6: refreshed; tools/merge_pmc_gemm.py)      # tools/run_pmc_gemm.sh: ONE GEMM of the block per process -> one row set per group
PMC_FILES = ("r6_pmc_traffic_attention_v8.csv", "r5_pmc_traffic_attention_v8.csv", "r4_pmc_traffic_attention_v8.csv", "r3_pmc_traffic_v5.csv", "r2_pmc_traffic_v3.csv", "r2_pmc_dominant_kernels.csv", "r1_pmc_dominant_kernels.csv")
PMC_KERNEL_OF_GROUP = {"attn_self": ("attn_fwd_kernel_v8", "attn_combine_kernel"), "attn_cross": ("attn_fwd_kernel_v2",)}


def pmc_traffic_bytes(group):
    """HBM/fabric bytes per launch of a kernel group from the committed rocprofv3 PMC passes (profiles/: FETCH_SIZE and WRITE_SIZE in KiB;
    FETCH_SIZE doubled per the gfx950 calibration in MI355X_MICROARCH.md, confirmed on the adaLN kernel: 2*FETCH = 4*L*C exactly).
    The GEMM groups come from per-shape passes (one GEMM of the block, with its epilogue, per profiled process; a group's kernels —
    the 256x256 launch and the 128x128 launch on its row remainder — are summed). None when no pass of that kernel is committed."""
    import csv
    try:
        rows = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", PMC_GROUP_FILE))) if r["group"] == group]
        if rows:
            return sum((2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0 for r in rows)
    except Exception:  # noqa: BLE001
        pass
    names = PMC_KERNEL_OF_GROUP.get(group)
    if not names:
        return None
    for f in PMC_FILES:
        try:
            rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", f))))
        except Exception:  # noqa: BLE001
            continue
        tot, hit = 0.0, False
        for n in names:
            for r in rows:
                if n in r["kernel"] and r.get("FETCH_SIZE") and r.get("WRITE_SIZE"):
                    tot += (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0
                    hit = True
                    break
        if hit:
            return tot
    return None


def _rand_vae(dev, version):
    from yume_amd import synth
    if version == "2.2":
        from yume_amd.wan23.modules.vae2_2 import Wan2_2_VAE as Wrap, WanVAE_
        cfg = synth.VAE_CFG_22
        with torch.device(dev):
            m = WanVAE_(dim=cfg["dim"], dec_dim=cfg["dec_dim"], z_dim=cfg["z_dim"], temperal_downsample=cfg["temperal_downsample"])
    else:
        from yume_amd.wan.modules.vae import WanVAE as Wrap, WanVAE_
        cfg = synth.VAE_CFG_21
        with torch.device(dev):
            m = WanVAE_(dim=cfg["dim"], z_dim=cfg["z_dim"], temperal_downsample=cfg["temperal_downsample"])
    g = torch.Generator(device=dev).manual_seed(5)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith("gamma"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g, device=dev))
            elif k.endswith("bias"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g, device=dev))
            else:
                p.copy_((torch.rand(p.shape, generator=g, device=dev) * 2 - 1) * (3.0 / p[0].numel()) ** 0.5)
    return Wrap(device=dev, model=m)


def _time_call(fn, reps=2):
    fn()                                                # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def vae_decode_rate(dev, z8):
    """Wan2.2 VAE decode of one chunk (8 latents [48,8,44,80] -> 29 frames 704x1280), random-init weights:
    the second half of BASELINE.json's metric ("VAE dec latents/s"). Not part of `value`. r5: the other VAE passes of the two pipelines ride
    along under `passes` with their own roofline fractions (FLOPs: SURVEY Appendix D / a17 — Wan2.2 17-frame encode 54.7 TFLOP, Wan2.1
    decode of 13 latents 218.6, Wan2.1 49-frame encode 130.2; the 14B pipeline re-encodes its history every chunk)."""
    vae = _rand_vae(dev, "2.2")
    dt, out = _time_call(lambda: vae.decode([z8])[0], reps=1)
    assert torch.isfinite(out).all()
    res = {"latents_per_s": z8.shape[1] / dt, "ms_per_chunk": dt * 1e3, "chunk": "8 latents 48x8x44x80 -> 29 frames 704x1280",
           "tflop_per_chunk": 485.04, "tflops": 485.04 / dt, "frac": 485.04 / dt / MFMA_BF16_PEAK_TFLOPS}
    passes = {}
    g = torch.Generator(device=dev).manual_seed(6)

    def add(name, fn, tflop, what, latents):
        try:
            d, o = _time_call(fn, reps=1)
            assert torch.isfinite(o).all()
            passes[name] = {"ms": d * 1e3, "tflop": tflop, "tflops": tflop / d, "frac": tflop / d / MFMA_BF16_PEAK_TFLOPS, "latents_per_s": latents / d,
                            "what": what}
        except Exception as e:  # noqa: BLE001 — a side measurement never hides the headline
            passes[name] = {"failed": f"{type(e).__name__}: {e}"}
    try:
        clip17 = torch.rand((3, 17, 704, 1280), generator=g, device=dev) * 2 - 1
        add("wan22_encode_17_frames", lambda: vae.encode([clip17])[0], 54.7, "Wan2.2 encode 3x17x704x1280 -> 48x5x44x80 (vae2_2.py:797-829)", 5)
        del vae, clip17
        torch.cuda.empty_cache()
        vae21 = _rand_vae(dev, "2.1")
        z13 = torch.randn((16, 13, 68, 120), generator=g, device=dev)
        add("wan21_decode_13_latents", lambda: vae21.decode([z13])[0], 218.6, "Wan2.1 decode 16x13x68x120 -> 3x49x544x960 (wan/modules/vae.py:544-568)", 13)
        clip49 = torch.rand((3, 49, 544, 960), generator=g, device=dev) * 2 - 1
        add("wan21_encode_49_frames", lambda: vae21.encode([clip49])[0], 130.2, "Wan2.1 encode 3x49x544x960 -> 16x13x68x120 (wan/modules/vae.py:516-542)", 13)
        del vae21
    except Exception as e:  # noqa: BLE001
        passes["failed"] = f"{type(e).__name__}: {e}"
    torch.cuda.empty_cache()
    res["passes"] = passes
    res["parity"] = ("every pass above is held to the fp32 device gold at exactly this size in tests/test_zy_vae_fullsize_gpu.py "
                     "(rel-L2 1.1e-2 ... 1.3e-2 decode, 6e-3 ... 1.1e-2 encode; tolerance 3e-2)")
    return res


def vae_cpu_baseline():
    """north_star: "VAE decode latents/sec, vs CPU reference". The oracle decoder (oracle/vae.py, fp32, pinned to the reference) on
    the host cores at a reduced spatial size — 2 latents [48,2,6,10] instead of [48,8,44,80] — with its FLOPs counted by torch's
    FlopCounterMode; the full-size rate is that FLOP rate divided by the 485.04/8 TFLOP one full-size latent costs."""
    from torch.utils.flop_counter import FlopCounterMode
    from oracle import vae as ovae
    from yume_amd import synth
    # 32 threads: the sample's convolutions are small, and with all 256 host threads of the GPU box torch's CPU conv3d spends its
    # time in thread hand-offs (measured 0.01 TFLOP/s there against 0.5 TFLOP/s on 8 cores)
    ncore = min(32, os.cpu_count() or 1)
    torch.set_num_threads(ncore)
    cfg = synth.VAE_CFG_22
    sd = synth.make_vae_state_dict(cfg, seed=5)
    z = torch.randn(48, 2, 6, 10, generator=torch.Generator().manual_seed(6))
    with FlopCounterMode(display=False) as fc:
        t0 = time.time()
        out = ovae.decode(sd, cfg, z)
        dt = time.time() - t0
    assert torch.isfinite(out).all()
    tf = fc.get_total_flops() / 1e12
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    return {"value": (tf / dt) / (485.04 / 8.0), "unit": "latents/s", "cores": ncore, "host_threads": os.cpu_count(), "kind": "port",
            "sample": f"oracle decode of 2 latents 48x2x6x10 ({tf:.2f} TFLOP, {dt:.2f} s = {tf / dt:.2f} TFLOP/s fp32), "
                      "scaled by FLOPs to the 704x1280 chunk (60.6 TFLOP per latent)"}


def port_vs_reference():
    """cpu_baseline.kind is "port" on the GPU box (no reference tree there): how the port's time relates to the REAL reference module's on
    the same host, measured in the build container by tools/port_vs_reference.py and committed (profiles/r5_cpu_port_vs_reference.json)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r5_cpu_port_vs_reference.json")))
        return {"reference_over_port": d["reference_over_port"], "reference_s": d["reference_s"], "port_s": d["port_s"], "threads": d["threads"],
                "outputs_rel_l2": d["outputs_rel_l2"], "measured": "build container (the only place the reference tree exists), tools/port_vs_reference.py: "
                + d["what"]}
    except Exception as e:  # noqa: BLE001
        return {"failed": str(e)}


def block_flops_5b(L, cfg, Lc=512):
    C, ffn = cfg["dim"], cfg["ffn_dim"]
    return 8 * L * C * C + 4 * L * L * C + 4 * L * C * C + 4 * Lc * C * C + 4 * L * Lc * C + 4 * L * C * ffn


def cpu_baseline(cfg, L, model):
    """Reference restatement on the host cores: one full-width block at the full sequence length (oracle/fullsize.py), and the
    same block on the same inputs through the device engine -> (cpu_baseline, parity). The thread count is swept on a 1/8-length
    slice first (all 256 hardware threads of the GPU box are 4x slower than 32-64: torch's CPU GEMM / SDPA lose their time in
    thread hand-offs), the full-L block runs at the best setting and `cores` is that thread count."""
    from oracle import fullsize
    ncpu = os.cpu_count() or 1
    cands = sorted({t for t in (8, 16, 32, 64, 128, 256) if t <= ncpu} | {min(ncpu, 8)})
    Ls = max(512, L // 8)
    small = fullsize.make_block_case(cfg, "wan23", Ls, seed=1)
    torch.set_num_threads(cands[0])
    fullsize.run_block_oracle(small)                                   # page in / warm the allocator, untimed
    sweep = {}
    for t in cands:
        torch.set_num_threads(t)
        _, ds = fullsize.run_block_oracle(small)
        sweep[t] = block_flops_5b(Ls, cfg) / ds / 1e12
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    case = fullsize.make_block_case(cfg, "wan23", L, seed=0)
    want, dt = fullsize.run_block_oracle(case)
    # (not back to every hardware thread: the host side of the engine is small tensor ops, and with the whole-step oracle running as a
    # subprocess next to the side workloads a 256-thread OpenMP team in THIS process is oversubscription — measured: seconds per first use
    # of a new clip geometry in the long-video workload)
    torch.set_num_threads(min(16, ncpu))
    tf = block_flops_5b(L, cfg) / dt / 1e12
    base = {"value": 1.0 / (dt * cfg["num_layers"]), "unit": "denoise-steps/sec", "cores": best, "host_threads": ncpu, "kind": "port",
            "sample": f"1 of {cfg['num_layers']} DiT blocks at L={L} (fp32, {dt:.2f} s = {tf:.2f} TFLOP/s on {best} of {ncpu} host threads), "
                      f"extrapolated x{cfg['num_layers']}; embed/head excluded; thread sweep on an L={Ls} block (TFLOP/s): "
                      + ", ".join(f"{t}: {v:.2f}" for t, v in sweep.items())
                      + "; kind 'port' = oracle/dit.py (restatement pinned to the reference): the GPU box has no reference tree to execute",
            "port_vs_reference": port_vs_reference()}
    # device leg: block 0 of the benchmarked model temporarily holds the case's weights
    blk = model.blocks[0]
    saved = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    blk.load_state_dict({k[len("blocks.0."):]: v for k, v in case["sd"].items()})
    try:
        rope_cs = torch.stack([case["rope"].real, case["rope"].imag], dim=-1).to(torch.float32)
        dev = blk.modulation.device
        got = model.engine.block_forward(0, case["x"].to(dev), case["e6"].to(dev), rope_cs.to(dev), case["ctx"].to(dev)).cpu()
    finally:
        blk.load_state_dict(saved)
    par = fullsize.parity(got, want)
    upd = fullsize.parity(got - case["x"], want - case["x"])
    par.update(update_rel_l2=upd["rel_l2"], tolerance="rel_l2 <= 1e-2 (bf16 MFMA path vs fp32 reference restatement)",
               what=f"one live 5B block at L={L}: device engine vs CPU oracle on identical inputs (weights in the model's dtype)")
    return base, par


def kernel_rooflines(prof, steps, ms_per_step, L, cfg, Lc=512):
    """per kernel group: launches and average launch time from the HIP events recorded inside the timed steps, algorithmic work
    per launch, achieved rate against the roofline that bounds it. Sorted by share of the step, largest first."""
    C, Fd, H = cfg["dim"], cfg["ffn_dim"], cfg["num_heads"]
    D = C // H
    work = {   # group -> (bound, algorithmic flop or bytes per launch, description)
        "attn_self": ("mfma", 4.0 * L * L * D * H, f"self-attention {L}x{L}x{H} heads, d={D}: attn_fwd_kernel_v8 (persistent workgroups, one continuous K/V^T "
                      "stream; + attn_combine_kernel for the key-range pieces of the last query blocks, timed together as one yume_attn_fwd_ws call)"),
        "attn_cross": ("mfma", 4.0 * L * Lc * D * H, f"cross-attention {L}x{Lc}x{H} heads: attn_fwd_kernel_v2"),
        "gemm_qkv": ("mfma", 2.0 * L * 3 * C * C, f"QKV GEMM {L}x{3 * C}x{C}, transposed-V epilogue (gemm_w4_kernel + gemm128_kernel on the row remainder)"),
        "gemm_o": ("mfma", 2.0 * L * C * C, f"o-proj GEMM {L}x{C}x{C}, gate*y + residual epilogue"),
        "gemm_cross_q": ("mfma", 2.0 * L * C * C, f"cross q GEMM {L}x{C}x{C}"),
        "gemm_cross_o": ("mfma", 2.0 * L * C * C, f"cross o-proj GEMM {L}x{C}x{C}, residual epilogue"),
        "gemm_ffn0": ("mfma", 2.0 * L * Fd * C, f"ffn.0 GEMM {L}x{Fd}x{C} + bias + GELU (gemm_w4_kernel + gemm128_kernel on the row remainder)"),
        "gemm_ffn2": ("mfma", 2.0 * L * C * Fd, f"ffn.2 GEMM {L}x{C}x{Fd}, gate*y + residual epilogue (gemm_w4_kernel with the split-K tail: 256 whole tiles + 188 tiles as heads / tails)"),
        "adaln": ("hbm", 6.0 * L * C, "LayerNorm + modulate: 4*L*C read + 2*L*C written"),
        # two launches per block share the group: q|k with RoPE (2C columns read + written in place, + the 512-byte fp32 cos / sin row of
        # the token) and the cross-attention q (C columns): the figure is their mean, as the launch time is
        "rmsnorm_rope": ("hbm", (4.0 * 2 * C * L + 512.0 * L + 4.0 * C * L) / 2, "RMSNorm (+RoPE) in place on q|k / cross q (2 B read + 2 B written per "
                         "element, + the token's RoPE row on q|k): mean of the block's two launches"),
    }
    out = []
    for name, evs in prof.items():
        if not evs:
            continue
        tot = sum(a.elapsed_time(b) for a, b in evs)
        avg = tot / len(evs)
        bound, w, desc = work.get(name, ("mfma", None, name))
        r = {"group": name, "bound": bound, "kernel": desc, "launch_ms": avg, "launches_timed": len(evs),
             "ms_per_step": tot / steps, "share_of_step": tot / steps / ms_per_step}
        if w is not None and avg > 0:
            if bound == "mfma":
                ach = w / (avg * 1e-3) / 1e12
                r.update(achieved=ach, peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / MFMA_BF16_PEAK_TFLOPS)
            else:
                ach = w / (avg * 1e-3) / 1e9
                r.update(achieved=ach, peak=8000.0, unit="GB/s", frac=ach / 8000.0)
        r["traffic"] = pmc_traffic_bytes(name)
        out.append(r)
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


def _coll_dev(dev):
    """device of the small timing / checksum collectives: the GPU under RCCL, the host under gloo (YUME_BENCH_SHARE_GPU test mode)."""
    return torch.device("cpu") if dist.is_initialized() and dist.get_backend() == "gloo" else torch.device(dev)


def _timed(fn, world):
    """barrier + synchronize on both sides, max over ranks (the driver's contract)."""
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=_coll_dev("cuda"))
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    return r, float(dt.item())


def _build_5b(args, rank, dev):
    from yume_amd import distributed as ydist
    from yume_amd import synth
    from yume_amd.wan23.modules.model import WanModel
    cfg = dict(synth.CFG_5B)
    if args.layers:
        cfg["num_layers"] = args.layers
    with torch.device(dev):
        model = WanModel(**cfg)
    if rank == 0:
        synth.randomize_module_(model, seed=0)
    model = model.to(torch.bfloat16).eval().requires_grad_(False)   # sample_5b.py:1241 casts the transformer to bf16
    n_bcast = ydist.broadcast_module_(model, src=0)
    return cfg, model, n_bcast


def bench_tts(args, rank, world, dev, built=None, emit=True):
    """BASELINE configs[3]: Yume-5B-720P SDE/TTS sampling (fastvideo/sample/sample_tts.py:694-868: eta 0.3, time_travel_step 2,
    interval 2 -> 74 model forwards per 50-step chunk), independent prompts sharded one per GPU (`index = (step-1)*world + rank`).
    A step = one SAMPLER step of that loop (its forward, its look-ahead forwards and updates); the window [warmup, warmup+steps)
    of the 50-step schedule is timed."""
    from yume_amd import framepack, sampling, synth
    cfg, model, n_bcast = built if built is not None else _build_5b(args, rank, dev)
    F, H, W, lfz, S, shift = 13, 44, 80, 8, 50, 7.0
    plan = framepack.pack_plan(F, H, W, lfz)
    g = torch.Generator(device=dev).manual_seed(3000 + rank)
    hist = torch.randn((48, F - lfz, H, W), generator=g, device=dev)
    latent = torch.cat([hist, torch.randn((48, lfz, H, W), generator=g, device=dev)], dim=1)
    ctx = [torch.randn((77, 4096), generator=g, device=dev)]
    sig = synth.sampling_sigmas(S, shift)
    vel = sampling.make_velocity_5b(model, ctx, plan.seq_len, plan.n_hist_tok, plan.n_new_tok, sig, lfz)
    histf = sampling.clean_history(hist)
    w0, w1 = args.warmup, min(S, args.warmup + args.steps)
    latent, cp = sampling.sde_tts_chunk(vel, latent, sig, lfz, histf, generator=g, i0=0, i1=w0, return_state=True)
    (latent, cp), dt = _timed(lambda: sampling.sde_tts_chunk(vel, latent, sig, lfz, histf, generator=g, i0=w0, i1=w1, current_pred=cp,
                                                             return_state=True), world)
    assert torch.isfinite(latent).all()
    nfw = sampling.tts_forward_count(S, i0=w0, i1=w1)
    res = None
    if rank == 0:
        k = w1 - w0
        res = ({"metric": "SDE/TTS sampler-steps/sec (Yume-5B 720P, 33-frame latent)", "value": world * k / dt, "unit": "sampler-steps/sec",
                          "n_gpus": world, "steps": k, "warmup": w0, "ms_per_step": dt / k * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": "Yume-5B-720P random-init, SDE (eta 0.3) + time-travel (step 2, interval 2) sampling of one "
                                                 "33-frame 704x1280 chunk, L=9460, one prompt per GPU", "tokens": plan.seq_len,
                                     "num_layers": cfg["num_layers"], "parallelism": f"dp{world} (independent prompts, replicated weights)"},
                          "model_forwards_timed": nfw, "forwards_per_s": world * nfw / dt, "forwards_per_50_step_chunk": sampling.tts_forward_count(S),
                          "ms_per_forward": dt / nfw * 1e3, "weight_broadcast_collectives": n_bcast,
                          "model_tflop_per_forward": flops_fwd_5b(plan.seq_len, n=cfg["num_layers"]) / 1e12,
                          "model_tflops_per_gpu": flops_fwd_5b(plan.seq_len, n=cfg["num_layers"]) / 1e12 * nfw / dt})
        if emit:
            print(json.dumps(res), flush=True)
    if world > 1 and emit:
        dist.destroy_process_group()
    return res


def bench_longvideo(args, rank, world, dev, built=None, emit=True):
    """BASELINE configs[4]: the FramePack long-video loop of fastvideo/sample/sample_5b.py:920-1097 — VAE encode of the conditioning
    clip, then `--chunks` (8) chunks of 2 s: `--steps` Euler steps each on [history | 8 noisy latents] (history grows by 8 latent
    frames per chunk: L 9460 ... 12545), VAE decode of the 8 new latents after every chunk. Everything is inside the timed region
    (the first chunk's encode included); one independent video per GPU. value = denoise steps of all ranks / time."""
    from yume_amd import framepack, sampling, synth
    from yume_amd.wan23.modules.vae2_2 import Wan2_2_VAE, WanVAE_
    cfg, model, n_bcast = built if built is not None else _build_5b(args, rank, dev)
    H, W, lfz, shift = 44, 80, 8, 7.0
    vcfg = synth.VAE_CFG_22
    with torch.device(dev):
        vm = WanVAE_(dim=vcfg["dim"], dec_dim=vcfg["dec_dim"], z_dim=vcfg["z_dim"], temperal_downsample=vcfg["temperal_downsample"])
    vm.load_state_dict(synth.make_vae_state_dict(vcfg, seed=5, device=dev), strict=True)
    vae = Wan2_2_VAE(device=dev, model=vm)
    g = torch.Generator(device=dev).manual_seed(4000 + rank)
    clip = torch.rand((3, 17, 704, 1280), generator=g, device=dev) * 2 - 1         # 17 frames -> 5 history latents
    ctxs = [torch.randn((77, 4096), generator=g, device=dev) for _ in range(args.chunks)]
    Ls, dec_ms = [], []

    def run(n_chunks, steps):
        hist = vae.encode([clip])[0]
        for k in range(n_chunks):
            plan = framepack.pack_plan(hist.shape[1] + lfz, H, W, lfz)
            Ls.append(plan.seq_len)
            hist, vids = sampling.long_video_5b(model, vae, hist, ctxs[k:k + 1], steps, shift, lfz, generator=g, decode=True)
            assert torch.isfinite(vids[0]).all()
        return hist
    run(1, max(1, args.warmup))
    Ls.clear()
    hist, dt = _timed(lambda: run(args.chunks, args.steps), world)
    # where a chunk's time goes (diagnostic pass behind the timed region, host clock with a synchronise between the parts)
    parts = {}
    if rank == 0:
        def lap(name, fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            parts[name] = (time.perf_counter() - t0) * 1e3
            return r
        h0 = lap("vae_encode_17_frames_ms", lambda: vae.encode([clip])[0])
        h1, _ = lap("chunk0_denoise_ms", lambda: sampling.long_video_5b(model, None, h0, ctxs[0:1], args.steps, shift, lfz, generator=g, decode=False))
        lap("chunk0_vae_decode_ms", lambda: vae.decode([h1[:, -lfz:]]))
    res = None
    if rank == 0:
        nsteps = args.chunks * args.steps
        tf = sum(flops_fwd_5b(l, n=cfg["num_layers"]) for l in Ls) / 1e12 * args.steps + args.chunks * 485.04 + 54.7
        res = ({"metric": "denoise-steps/sec (Yume-5B FramePack long video, VAE encode/decode per chunk in the timed region)",
                          "value": world * nsteps / dt, "unit": "denoise-steps/sec", "n_gpus": world, "steps": nsteps, "warmup": args.warmup,
                          "ms_per_step": dt / nsteps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                          "data": "synthetic",
                          "config": {"workload": f"Yume-5B-720P random-init, {args.chunks} x 2 s chunks of 704x1280, {args.steps} Euler steps per chunk "
                                                 "(the reference uses 50), 17-frame conditioning clip, Wan2.2 VAE encode + per-chunk decode",
                                     "tokens_per_chunk": Ls, "num_layers": cfg["num_layers"], "parallelism": f"dp{world} (independent videos, replicated weights)"},
                          "latents_per_s": world * args.chunks * lfz / dt, "final_history_latents": int(hist.shape[1]),
                          "weight_broadcast_collectives": n_bcast,
                          "model_tflop_timed": tf, "model_tflops_per_gpu": tf / dt, "parts_of_one_chunk": parts,
                          "model_tflop_note": "DiT forwards of every chunk at its own L + 485.04 per chunk decode + 54.7 for the 17-frame encode",
                          "read_as": f"a {args.steps}-step-per-chunk run of the configs[4] loop (the reference samples 50 steps per chunk: there the denoise "
                                     "steps are 93 % of a chunk and the rate approaches the headline's); NOT configs[4]'s own number unless --steps 50"})
        if emit:
            print(json.dumps(res), flush=True)
    if world > 1 and emit:
        dist.destroy_process_group()
    return res


def bench_14b(args, rank, world, dev, emit=True):
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
    tmax = torch.tensor([dt], dtype=torch.float64, device=_coll_dev(dev))
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = float(tmax.item())
    res = None
    if rank == 0:
        ms = tmax / args.steps * 1e3
        tf = 2 * 1319.3 * (cfg["num_layers"] / 40.0)
        res = ({"metric": "denoise-steps/sec (Yume-I2V-14B 540P, 65-frame latent, CFG)", "value": world * args.steps / tmax,
                          "unit": "denoise-steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": "Yume-I2V-14B-540P random-init, latent 16x17x68x120 + y, FramePack lfz=9, L=27810, CFG 5.0 "
                                                 "(2 forwards/step), 50-step shift-3 schedule", "num_layers": cfg["num_layers"], "tokens": L},
                          "model_tflop_per_step": tf, "model_tflops_per_gpu": tf / (ms * 1e-3),
                          "peak_mem_gb": torch.cuda.max_memory_allocated() / 1e9,
                          "parity": "this exact configuration (L = 27810, 40 blocks, CFG 5.0) is held to the fp32 device gold in "
                                    "tests/test_zz_full_step_gpu.py::test_full_depth_14b_cfg_step_at_the_benchmarked_length_vs_device_gold "
                                    "(cond 5.0e-3, uncond 4.8e-3, guided velocity 1.3e-2, updated latent 2.1e-4; tolerance 3e-2 / 4e-2 / 3e-3)"})
        if emit:
            print(json.dumps(res), flush=True)
    if world > 1 and emit:
        dist.destroy_process_group()
    return res


def other_workloads(args, dev, built):
    """BASELINE configs[2], [3], [4] in the same process, behind the headline's timed region (each has its own barrier-bracketed timed
    region; `python bench.py --workload X` runs one of them alone, at any N): {name: that workload's JSON object}."""
    import copy
    res = {}
    for name, fn, kw, ov in (("tts", bench_tts, {"built": built}, dict(steps=4, warmup=2)),
                             ("longvideo", bench_longvideo, {"built": built}, dict(steps=8, warmup=1, chunks=8)),
                             ("14b", bench_14b, {}, dict(steps=3, warmup=1))):
        a = copy.copy(args)
        for k, v in ov.items():
            setattr(a, k, v)
        try:
            t0 = time.perf_counter()
            res[name] = fn(a, 0, 1, dev, emit=False, **kw)
            res[name]["wall_s_incl_build"] = time.perf_counter() - t0
        except Exception as e:  # noqa: BLE001 — never hide the headline behind a side workload
            res[name] = {"failed": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    return res


def full_step_parity(job, model, cfg):
    """the WHOLE denoise step of BASELINE configs[1] — 30 live blocks + head at L = 9460 through WanModel.forward and
    the Euler update of sample_5b.py:985-990 — on the device against oracle.dit.forward_wan23 (fp32, run as the subprocess `job` on the
    host cores: oracle/step_job.py) on identical latent / timestep / text-embedding inputs and identical (hashed synthetic) weights.
    -> (parity dict, cpu_baseline dict: the oracle's measured whole-step time instead of one block x 30)."""
    from oracle import step_job
    from yume_amd import synth
    proc, path = job
    synth.fill_module_hashed_(model, cfg, "wan23", step_job.SEED)            # the benchmarked module now holds the case's weights (bf16)
    pred = step_job.device_forward("5b", model, "cond").cpu()
    ref = step_job.finish_job(proc, path)
    for f in (path, path + ".log"):
        if os.path.exists(f):
            os.remove(f)
    lat, i = step_job.make_inputs("5b")["latent"], step_job.CASES["5b"]["i"]
    p = step_job.stats(pred, ref["pred"])
    u = step_job.stats(step_job.euler("5b", lat, pred, i), step_job.euler("5b", lat, ref["pred"], i))
    par = {"pred_rel_l2": p["rel_l2"], "pred_max_abs": p["max_abs"], "pred_rms": p["ref_rms"], "latent_rel_l2": u["rel_l2"],
           "latent_max_abs": u["max_abs"], "tolerance": "pred rel_l2 <= 3e-2, updated latent rel_l2 <= 2e-3 (DESIGN.md 5)",
           "what": "one whole denoise step of configs[1] (30 live blocks + head, L=9460, sigma index 10 of 50, shift 7) + Euler update: device "
                   "(bf16 weights, bf16 MFMA) vs CPU oracle (fp32) on identical inputs"}
    tf = flops_fwd_5b(9460, n=cfg["num_layers"]) / 1e12
    base = {"value": 1.0 / ref["seconds"], "unit": "denoise-steps/sec", "cores": ref["threads"], "host_threads": ref["host_threads"], "kind": "port",
            "sample": f"ONE WHOLE denoise step (30 blocks + embeddings + head at L=9460, fp32): {ref['seconds']:.1f} s = {tf / ref['seconds']:.2f} TFLOP/s on "
                      f"{ref['threads']} of {ref['host_threads']} host threads (+ {ref['gen_seconds']:.1f} s generating the weights, not counted), measured while "
                      "the GPU ran the side workloads; kind 'port' = oracle/dit.py (restatement pinned to the reference): the GPU box has no "
                      "reference tree to execute",
            "port_vs_reference": port_vs_reference()}
    return par, base


def _self_launch(args, share):
    """`python bench.py --gpus N` outside torchrun: become the N-rank job (one process per GPU over RCCL, the launch the
    reference's scripts use — torchrun --nproc_per_node N fastvideo/sample/sample_5b.py, scripts/inference/sample_5b.sh; one
    process per GPU at sample_5b.py:1124-1134). Refuses by name when the node has fewer than N GPUs."""
    n = args.gpus
    have = torch.cuda.device_count()
    if have < n and not share:
        sys.exit(f"bench.py: --gpus {n} requested but this node shows {have} GPU(s): refusing to run {n} ranks on fewer devices "
                 "(a 1-rank run would print n_gpus: 1, not what was asked)")
    # --standalone: the launcher's own c10d rendezvous picks a free port itself (no bind / close / reuse race of a pre-probed port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node to use; default: the launcher's WORLD_SIZE, else 1")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=0, help="debug only: override num_layers (result is then NOT the named config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the (untimed-in-value) VAE decode measurement")
    ap.add_argument("--sp", action="store_true",
                    help="5b only, not the headline: ONE chain split over the N ranks (Ulysses sequence parallelism, "
                         "SURVEY 8(f).2); value is that chain's steps/s, scaling 'strong'")
    ap.add_argument("--workload", default="5b", choices=["5b", "14b", "tts", "longvideo"],
                    help="5b = BASELINE configs[1] (the headline metric, default); 14b = configs[2] (Yume-I2V-14B-540P, 65-frame clip, CFG); "
                         "tts = configs[3] (SDE/TTS sampling, sampler steps); longvideo = configs[4] (FramePack chunks with VAE encode/decode)")
    ap.add_argument("--chunks", type=int, default=8, help="longvideo only: number of 2 s chunks (BASELINE: 8)")
    ap.add_argument("--no-full-parity", action="store_true",
                    help="skip the whole-step parity leg (30 blocks + head at L = 9460 against the CPU oracle: ~3-4 min of host time, "
                         "run as a subprocess behind the timed region)")
    ap.add_argument("--no-workloads", action="store_true",
                    help="5b only: do not attach BASELINE configs[2] / [3] / [4] (14B CFG, SDE/TTS, long video) as `workloads` to the JSON line")
    args = ap.parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))          # `torchrun --nproc-per-node N bench.py` alone means N ranks

    share = os.environ.get("YUME_BENCH_SHARE_GPU", "0") == "1"   # test mode only: every rank on cuda:0, gloo instead of RCCL
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _self_launch(args, share)                                 # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if share else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; pass --gpus {world} "
                 f"(or run `python bench.py --gpus {args.gpus}` alone: it launches its own ranks)")
    if not share and torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks need {world} GPUs, this node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        # N ranks share one host: each keeps a small OpenMP / intra-op team (the engine's host side is small tensor ops; 8 ranks x torch's
        # default of every hardware thread oversubscribes the box, and rank 0's CPU legs start only after the other ranks have left)
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 8) // world)))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")              # collectives staged through the host (yume_amd.distributed)
        else:
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
    if args.workload == "tts":
        return bench_tts(args, rank, world, dev)
    if args.workload == "longvideo":
        return bench_longvideo(args, rank, world, dev)
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

    # HIP-event brackets around every kernel call cost queue time (two marker packets per call, ~330 calls per step: measured ~3 ms of
    # inter-kernel gaps per step in the rocprofv3 trace). So: the last warmup step is fully instrumented and names the dominant kernel
    # group; the TIMED steps bracket only that group (`roofline`); the other groups (`roofline_all`) come from fully instrumented extra
    # steps behind the timed region.
    dominant = "attn_self"
    for i in range(args.warmup):
        if i == args.warmup - 1:
            model.engine.prof, model.engine.prof_only = {}, None
        latent = step(i, latent)
    if model.engine.prof:
        torch.cuda.synchronize()
        dominant = max(model.engine.prof.items(), key=lambda kv: sum(a.elapsed_time(b) for a, b in kv[1]))[0]
    model.engine.prof, model.engine.prof_only = {}, {dominant}
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
    prof, model.engine.prof, model.engine.prof_only = model.engine.prof, None, None
    n_extra, prof_all = 2, {}
    if rank == 0:
        lat_x = latent
        model.engine.prof = prof_all
        for i in range(n_extra):
            lat_x = step(args.warmup + args.steps + i, lat_x)
        torch.cuda.synchronize()
        model.engine.prof = None
        del lat_x
    assert torch.isfinite(latent).all(), "non-finite latents"
    # per-box calibration, right behind the timed steps (same thermal / power state): what this chip sustains under a pure MFMA load and
    # on one fixed reference launch of the product GEMM (yume_amd/calibrate.py)
    calib = None
    if rank == 0:
        try:
            from yume_amd import calibrate
            cm, cg = calibrate.mfma_sustained(dev), calibrate.gemm_reference(dev)
            calib = {"mfma_sustained_tflops": cm["tflops"], "mfma_clock_ghz": cm["clock_ghz"], "s_memtime_ghz": cm["s_memtime_ghz"],
                     "gemm_8192_tflops": cg["tflops"], "gemm_8192_ms": cg["ms_per_launch"], "kernel": cm["kernel"],
                     "note": "measured in this process right behind the timed steps; frac_of_sustained = achieved / mfma_sustained_tflops "
                             "(frac stays achieved / the nominal 2500 TFLOP/s = 2.4 GHz x 32 clocks per 32x32x16 MFMA x 4 SIMDs x 256 CUs)"}
        except Exception as e:  # noqa: BLE001
            calib = {"failed": f"{type(e).__name__}: {e}"}
    # results of every chain are gathered at chunk end (the only other collective of the run)
    checks = ydist.gather_scalars(float(latent[:, -lfz:].double().abs().mean()), device=_coll_dev(dev))

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
    # VERDICT r5 #6 (the last block's history rows supply K / V only): reported beside the headline like the cached context, never as `value`
    trimmed_ms = None
    if rank == 0 and not sp:
        model.engine.trim_last_block = True
        lat2 = step(0, latent)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for i in range(1, 1 + max(2, min(args.steps, 5))):
            lat2 = step(i, lat2)
        torch.cuda.synchronize()
        trimmed_ms = (time.perf_counter() - tc) / max(2, min(args.steps, 5)) * 1e3
        model.engine.trim_last_block = False
        del lat2

    vae_res = None
    if not args.no_vae and rank == 0:
        vae_res = vae_decode_rate(dev, latent[:, -lfz:].float())

    tmax = torch.tensor([dt], dtype=torch.float64, device=_coll_dev(dev))
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = float(tmax.item())
    threads_per_rank = None
    if world > 1:
        threads_per_rank = [int(v) for v in ydist.gather_scalars(float(torch.get_num_threads()), device=_coll_dev(dev))]
        rank_log = os.environ.get("YUME_BENCH_RANK_LOG")          # tests: every rank leaves a record of when its GPU work was over
        if rank_log:
            with open(os.path.join(rank_log, f"rank{rank}.json"), "w") as f:
                json.dump({"rank": rank, "pid": os.getpid(), "threads": torch.get_num_threads(), "gpu_work_done_at": time.time()}, f)
        dist.barrier()
        dist.destroy_process_group()      # the last collective is behind us: the other ranks leave, rank 0 goes on to the host-side legs alone
    cpu_legs_started_at = time.time()

    if rank == 0:
        ms_per_step = tmax / args.steps * 1e3
        rl_dom = kernel_rooflines(prof, args.steps, ms_per_step, L, cfg)           # the dominant group, inside the timed steps
        rl_rest = [r for r in kernel_rooflines(prof_all, n_extra, ms_per_step, L, cfg) if r["group"] != dominant]
        rl_all = rl_dom + rl_rest
        if calib and calib.get("mfma_sustained_tflops"):
            for r in rl_all:
                if r.get("bound") == "mfma" and r.get("achieved"):
                    r["frac_of_sustained"] = r["achieved"] / calib["mfma_sustained_tflops"]
        out = {
            "metric": "denoise-steps/sec (Yume-5B 720P, 33-frame latent)",
            "value": (1 if sp else world) * args.steps / tmax, "unit": "denoise-steps/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if sp else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Yume-5B-720P random-init, 33-frame 704x1280 clip (latent 48x13x44x80, FramePack "
                                   "lfz=8, L=9460), ODE Euler steps of a 50-step shift-7 schedule, no CFG, one chain per GPU",
                       "num_layers": cfg["num_layers"], "tokens": L, "parallelism": (f"sp{world} (one chain, Ulysses all-to-all, replicated weights)" if sp
                                       else f"dp{world} (independent chains, replicated weights)")},
            "calibration": calib,
            "chain_checksums": checks, "weight_broadcast_collectives": n_bcast,
            "vae_decode": vae_res,
            "cached_context_ms_per_step": cached_ms,
            "trimmed_last_block_ms_per_step": trimmed_ms,
            "host_threads_per_rank": threads_per_rank, "cpu_legs_started_at": cpu_legs_started_at,
            "model_tflop_per_step": flops_fwd_5b(L, n=cfg["num_layers"]) / 1e12,
            "model_tflops_per_gpu": flops_fwd_5b(L, n=cfg["num_layers"]) / 1e12 / (ms_per_step * 1e-3),
            "roofline": rl_all[0] if rl_all else None,          # the kernel group with the largest share of the step
            "roofline_all": rl_all[1:],
            "roofline_all_source": f"{n_extra} fully instrumented steps behind the timed region (HIP events around every kernel call); "
                                   "the timed steps bracket only the dominant group",
        }
        # cpu_baseline / parity are emitted by rank 0 at ANY world size (the CPU legs run on rank 0's host cores after the timed region;
        # the other ranks have nothing left to do and leave)
        job = None
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"], blk_par = cpu_baseline(cfg, L, model)
                out["parity"] = {"block": blk_par}
            except Exception as e:  # noqa: BLE001 — a baseline failure must not hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "denoise-steps/sec", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e}"}
            if vae_res is not None:
                try:
                    vae_res["cpu_baseline"] = vae_cpu_baseline()
                except Exception as e:  # noqa: BLE001
                    vae_res["cpu_baseline"] = {"value": None, "unit": "latents/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
            if not args.no_full_parity and not args.layers:
                try:                                  # the whole-step oracle: a subprocess on 32 host threads while the GPU runs the workloads below
                    import tempfile
                    from oracle import step_job
                    job_out = os.path.join(tempfile.gettempdir(), f"yume_bench_step_{os.getpid()}.pt")
                    job = (step_job.start_job("5b", "cond", job_out, threads=out["cpu_baseline"].get("cores") or 32), job_out)
                except Exception as e:  # noqa: BLE001
                    out.setdefault("parity", {})["full_step"] = {"failed": str(e)}
        if world == 1 and not args.no_workloads and not args.layers:
            out["workloads"] = other_workloads(args, dev, (cfg, model, n_bcast))
        if job is not None:
            try:
                out.setdefault("parity", {})["full_step"], full_base = full_step_parity(job, model, cfg)
                full_base["block_sample"] = out["cpu_baseline"]
                out["cpu_baseline"] = full_base
            except Exception as e:  # noqa: BLE001
                out.setdefault("parity", {})["full_step"] = {"failed": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
