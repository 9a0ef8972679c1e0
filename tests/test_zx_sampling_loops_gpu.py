"""Row N3: the PRODUCT's sampling loops — yume_amd/sampling.py, the code bench.py's `workloads.tts` / `workloads.longvideo`
time — driving the live 30-block 5B model (and the Wan2.2 VAE) on the device, VALUE-checked against the oracle loops of
oracle/sampler.py driving the device gold (oracle/step_job.oracle_forward(device="cuda"): oracle/dit.py on the GPU in fp32, proven
against the CPU oracle at this size in tests/test_zz_full_step_gpu.py; oracle/devgold.vae_*: oracle/vae.py likewise, proven in
tests/test_zy_vae_fullsize_gpu.py). Both chains run on their OWN results from the same start, so a figure is the difference the whole
loop accumulates: model error, update arithmetic, the order of the forwards, the history put in front of every step, the hand-over
between chunks.

  (a) sampling.sde_tts_chunk      fastvideo/sample/sample_tts.py:694-868 — a whole 6-step chunk (SDE eta 0.3, time travel step 2 /
                                  interval 2: look-aheads behind steps 0 and 2, the stale `current_pred` reuse behind step 4, the final
                                  step to sigma 0), full area (L = 9460), 8 forwards per side. The SDE noise is REPLAYED: the product draws
                                  from a device generator (its production path, sampling._sde_step), the oracle's `randn` draws the same
                                  shapes from a second device generator with the same seed — identical values in identical order.
  (b) sampling.long_video_5b      fastvideo/sample/sample_5b.py:920-1097 — Wan2_2_VAE.encode of a 17-frame 704x1280 clip, 2 chunks x 2
                                  Euler steps (L = 9460, then 11 420), the history hand-over between the chunks, Wan2_2_VAE.decode of each
                                  chunk's 8 new latents; vs devgold.vae_encode -> oracle.sampler.long_video_5b on the gold model ->
                                  devgold.vae_decode. Chunk noise replayed as above.
  (c) sampling.ode_chunk          sample_5b.py:960-1034 — a whole 10-step Euler chunk at full area through the product's loop (the r5 drift
                                  record tools/chain_drift.py re-implemented the loop; this one calls it).

Stated tolerances (bf16 device model vs fp32 gold, accumulated over the loop): final new latents rel-L2 <= 1e-2 for (a) and
(c); (b): encoded history <= 3e-2 (the VAE tolerance), new latents of each chunk <= 3e-2 (they inherit the history's VAE error through
the history tokens), decoded frames <= 5e-2 overall and per frame. Measured values are printed."""
import sys
import time

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)

from oracle import devgold  # noqa: E402
from oracle import sampler as osamp  # noqa: E402
from oracle import step_job  # noqa: E402
from yume_amd import sampling, synth  # noqa: E402

DEV = "cuda"
_STATE = {}


@pytest.fixture(scope="module")
def model():
    """the 30-block 5B device model (bf16, hashed weights) and the gold's fp32 weights (generated once, kept for the module's 22 gold forwards)."""
    m = step_job.build_device_model("5b", DEV)
    assert step_job.weights_agree("5b", m)
    _STATE["gold_weights"] = {}
    yield m
    _STATE.clear()
    del m
    torch.cuda.empty_cache()


class Counted:
    def __init__(self, fn):
        self.fn, self.n = fn, 0

    def __call__(self, *a):
        self.n += 1
        return self.fn(*a)


def _gold_transformer(name, ctx=None):
    def f(lat, i, which, k=None):
        c = None if ctx is None else ctx[k]
        return step_job.oracle_forward(name, which, latent=lat, i=i, device=DEV, ctx=c, weight_cache=_STATE["gold_weights"])[0]
    return Counted(f)


def _replay(seed, full_shape_tail=None):
    """oracle-side `randn(shape)`: the product's draws again — same device generator seed, same shapes, same order — on the host in fp32.
    full_shape_tail = lfz: the long-video script draws noise of the padded latent's shape and uses its last lfz frames (sample_5b.py:945-953),
    the product draws those frames only; the replay puts the product's draw where the script reads it."""
    g = torch.Generator(device=DEV).manual_seed(seed)

    def randn(shape):
        shape = tuple(shape)
        if full_shape_tail is None:
            return torch.randn(shape, generator=g, device=DEV, dtype=torch.float32).cpu()
        C, F, H, W = shape
        out = torch.zeros(shape)
        out[:, -full_shape_tail:] = torch.randn((C, full_shape_tail, H, W), generator=g, device=DEV, dtype=torch.float32).cpu()
        return out
    return randn


def test_sde_tts_chunk_on_the_device_vs_oracle_loop_on_device_gold(model):
    name = "5b_tts6"
    c, sg, plan = step_job.CASES[name], step_job.sigmas(name), step_job.seq_len(name)
    lfz, S = c["lfz"], c["steps"]
    assert plan.seq_len == 9460 and S == 6
    inp = step_job.make_inputs(name)
    lat0, hist = inp["latent"], inp["latent"][:, :-lfz]
    t0 = time.time()
    vel = Counted(sampling.make_velocity_5b(model, [inp["cond"].to(DEV)], plan.seq_len, plan.n_hist_tok, plan.n_new_tok, sg, lfz))
    got = sampling.sde_tts_chunk(vel, lat0.to(DEV), sg, lfz, sampling.clean_history(hist.to(DEV)),
                                 generator=torch.Generator(device=DEV).manual_seed(606)).cpu()
    td = time.time() - t0
    gold = _gold_transformer(name)
    want = osamp.tts(gold, lat0, lat0, None, sg, lfz, _replay(606), sde=True, cfg=False, renoise=False)
    assert vel.n == gold.n == sampling.tts_forward_count(S) == 8          # 6 steps + look-aheads behind steps 0 and 2
    assert got.shape == want.shape == lat0.shape and torch.isfinite(got).all()
    assert torch.equal(got[:, :-lfz], hist) and torch.equal(want[:, :-lfz], hist)                     # clean history in front, untouched
    s = step_job.stats(got[:, -lfz:], want[:, -lfz:])
    # how far the chunk moved the new latents from their start: the scale a wrong sign / slice / schedule error would show up at
    moved = step_job.stats(want[:, -lfz:], lat0[:, -lfz:])
    print(f"sde_tts_chunk, 6 sampler steps / 8 forwards, L=9460, noise replayed: final new latents rel-L2 {s['rel_l2']:.3e} max-abs {s['max_abs']:.3e} "
          f"(rms {s['ref_rms']:.3f}; the chunk moved the latents by {moved['rel_l2']:.2f} of their norm); device loop {td:.1f} s, test {time.time() - t0:.0f} s")
    assert s["rel_l2"] <= 1e-2


def test_ode_chunk_ten_steps_on_the_device_vs_oracle_loop_on_device_gold(model):
    name = "5b_ode10"
    c, sg, plan = step_job.CASES[name], step_job.sigmas(name), step_job.seq_len(name)
    lfz = c["lfz"]
    inp = step_job.make_inputs(name)
    lat0, hist = inp["latent"], inp["latent"][:, :-lfz]
    t0 = time.time()
    vel = Counted(sampling.make_velocity_5b(model, [inp["cond"].to(DEV)], plan.seq_len, plan.n_hist_tok, plan.n_new_tok, sg, lfz))
    got = sampling.ode_chunk(vel, lat0.to(DEV), sg, lfz, sampling.clean_history(hist.to(DEV))).cpu()
    gold = _gold_transformer(name)
    want = osamp.euler_5b(gold, lat0, lat0, sg, lfz)
    assert vel.n == gold.n == 10
    assert torch.equal(got[:, :-lfz], hist)
    s = step_job.stats(got[:, -lfz:], want[:, -lfz:])
    print(f"ode_chunk, the whole 10-step shift-7 schedule, L=9460: final new latents rel-L2 {s['rel_l2']:.3e} max-abs {s['max_abs']:.3e} "
          f"(rms {s['ref_rms']:.3f}); test {time.time() - t0:.0f} s")
    assert torch.isfinite(got).all() and s["rel_l2"] <= 1e-2


def test_long_video_loop_with_vae_on_the_device_vs_oracle_loop_on_device_gold(model):
    name, vseed, n_chunks = "5b_lv2", 61, 2
    c, sg = step_job.CASES[name], step_job.sigmas(name)
    lfz, steps, shift = c["lfz"], c["steps"], c["shift"]
    assert sg == [float(v) for v in synth.sampling_sigmas(steps, shift)]
    g = torch.Generator().manual_seed(62)
    clip = torch.rand(3, 17, 704, 1280, generator=g) * 2 - 1
    ctxs = [torch.randn(77, 4096, generator=g) for _ in range(n_chunks)]
    # ---- the product: Wan2_2_VAE.encode -> sampling.long_video_5b (-> Wan2_2_VAE.decode per chunk)
    from yume_amd.wan23.modules.vae2_2 import Wan2_2_VAE, WanVAE_
    vcfg = synth.VAE_CFG_22
    vm = WanVAE_(dim=vcfg["dim"], dec_dim=vcfg["dec_dim"], z_dim=vcfg["z_dim"], temperal_downsample=vcfg["temperal_downsample"])
    vm.load_state_dict(synth.make_vae_state_dict(vcfg, seed=vseed), strict=True)
    vae = Wan2_2_VAE(z_dim=vcfg["z_dim"], device=DEV, model=vm)
    t0 = time.time()
    h0 = vae.encode([clip.to(DEV)])[0]
    chunk_lat = []
    hist, vids = sampling.long_video_5b(model, vae, h0, [x.to(DEV) for x in ctxs], steps, shift, lfz,
                                        generator=torch.Generator(device=DEV).manual_seed(707), decode=True,
                                        on_chunk=lambda k, lat: chunk_lat.append(lat.cpu()))
    h0, hist, vids = h0.cpu(), hist.cpu(), [v.cpu() for v in vids]
    td = time.time() - t0
    # ---- the gold: devgold.vae_encode -> oracle.sampler.long_video_5b on the gold model -> devgold.vae_decode
    g0 = devgold.vae_encode("2.2", clip, vseed, DEV)
    gold = _gold_transformer(name, ctxs)
    ghist, gvids = osamp.long_video_5b(gold, lambda z: devgold.vae_decode("2.2", z, vseed, DEV), g0, n_chunks, sg, lfz, _replay(707, lfz))
    assert gold.n == n_chunks * steps
    # ---- structure of the hand-over (sample_5b.py:1043-1049,1093-1095): history grows by lfz frames per chunk, old frames untouched
    F0 = 5
    assert h0.shape == g0.shape == (48, F0, 44, 80)
    assert hist.shape == ghist.shape == (48, F0 + n_chunks * lfz, 44, 80)
    assert torch.equal(hist[:, :F0], h0) and torch.equal(chunk_lat[0], hist[:, :F0 + lfz]) and torch.equal(chunk_lat[1], hist)
    assert [step_job.seq_len(name, F0 + (k + 1) * lfz).seq_len for k in range(n_chunks)] == [9460, 11420]
    # ---- values
    e = step_job.stats(h0, g0)
    print(f"long_video_5b, 17-frame 704x1280 clip, {n_chunks} chunks x {steps} steps: encoded history rel-L2 {e['rel_l2']:.3e}; device loop {td:.1f} s")
    assert torch.isfinite(hist).all() and e["rel_l2"] <= 3e-2
    for k in range(n_chunks):
        sl = slice(F0 + k * lfz, F0 + (k + 1) * lfz)
        s = step_job.stats(hist[:, sl], ghist[:, sl])
        v = step_job.stats(vids[k], gvids[k])
        d = vids[k].double() - gvids[k].double()
        pf = [(d[:, t].norm() / gvids[k][:, t].double().norm().clamp_min(1e-30)).item() for t in range(gvids[k].shape[1])]
        print(f"  chunk {k}: new latents rel-L2 {s['rel_l2']:.3e} max-abs {s['max_abs']:.3e} (rms {s['ref_rms']:.3f}); decoded {tuple(vids[k].shape)} "
              f"rel-L2 {v['rel_l2']:.3e}, worst frame {max(pf):.3e}")
        assert vids[k].shape == gvids[k].shape and torch.isfinite(vids[k]).all()
        assert s["rel_l2"] <= 3e-2 and v["rel_l2"] <= 5e-2 and max(pf) <= 5e-2
    print(f"  test {time.time() - t0:.0f} s")


def test_trimmed_last_block_at_full_size_returns_the_same_velocity(model):
    """engine.trim_last_block on the full 30-block 5B model at L = 9460: the last block computes queries / o / cross-attention /
    FFN for the 7040 new rows only (2420 history rows supply K / V). Same kernels on row-offset views; what may differ is which tile
    kernel a row lands in (the 256-row main launch or the 128-row-tile remainder launch: 7040 rows split differently from 9460) and which
    query blocks take a key-range split — fp32 sums in another order. So the claim is stated as measured: bits are equal at the tiny sizes
    of tests/test_dit_gpu.py, here the difference is rel-L2 3.7e-5 (first run), 1/150 of the velocity's own bf16-vs-fp32 error (5.8e-3);
    asserted <= 1e-4."""
    name = "5b"
    base = step_job.device_forward(name, model, "cond").clone()
    model.engine.trim_last_block = True
    try:
        got = step_job.device_forward(name, model, "cond").clone()
    finally:
        model.engine.trim_last_block = False
    s = step_job.stats(got, base)
    print(f"trim_last_block at L=9460: bit-identical {torch.equal(got, base)}; rel-L2 {s['rel_l2']:.3e} max-abs {s['max_abs']:.3e}")
    assert got.shape == base.shape and s["rel_l2"] <= 1e-4


def test_stream_capture_and_replay_of_one_denoise_step(model):
    """VERDICT r5 missing #4: one whole denoise step (30 blocks + head, ~330 launches of libyume_hip, the persistent attention kernel's ticket
    counters included) captured into a HIP graph and replayed — the launch form of the 4-step interactive case (scripts/inference/
    sample_5b.sh:18, webapp_single_gpu.py:806). The counter workspace is registered OUTSIDE the capture (ops.ensure_counters refuses inside);
    a replay on new latents has to return the bits of the eager step on those latents. The ms of both forms are printed (at this size the
    step is GPU-bound: the graph saves host time, not GPU time)."""
    from yume_amd import ops
    name = "5b"
    c, sg, plan = step_job.CASES[name], step_job.sigmas(name), step_job.seq_len(name)
    lfz = c["lfz"]
    inp = step_job.make_inputs(name)
    lat = inp["latent"].to(DEV).clone()
    ctx = [inp["cond"].to(DEV)]
    t = torch.cat([torch.zeros(plan.n_hist_tok, dtype=torch.float64), torch.full((plan.n_new_tok,), sg[c["i"]] * 1000.0, dtype=torch.float64)]).unsqueeze(0).to(DEV)
    ops.ensure_counters(torch.device(DEV, torch.cuda.current_device()))

    def fwd():
        return model([lat], t=t, context=ctx, seq_len=plan.seq_len, latent_frame_zero=lfz, flag=True)[0]
    base = fwd().clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fwd()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, base)
    lat.copy_(torch.randn(lat.shape, generator=torch.Generator().manual_seed(5)).to(DEV))
    graph.replay()
    got = out.clone()
    want = fwd().clone()
    assert torch.isfinite(got).all() and not torch.equal(got, base)
    assert torch.equal(got, want)

    def ms(fn, n=5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    e, r = ms(fwd), ms(graph.replay)
    print(f"one 5B denoise forward at L=9460: eager {e:.2f} ms, hipGraph replay {r:.2f} ms (bit-identical results)")
    del graph
