"""Host logic of the denoise loops (yume_amd/sampling.py) against the independent restatement of the reference scripts
(oracle/sampler.py), with a deterministic stand-in velocity field instead of the DiT (CPU, no GPU needed)."""
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)
from oracle import sampler as osamp  # noqa: E402
from yume_amd import sampling  # noqa: E402


def fake_field(seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(6, 6, generator=g) * 0.3

    def f(latent, i, which):
        s = 1.0 if which == "cond" else 0.7
        return torch.tanh(torch.einsum("cd,dfhw->cfhw", w, latent)) * s + 0.01 * i
    return f


class Replay:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)

    def __call__(self, shape):
        return torch.randn(shape, generator=self.g)


@pytest.mark.parametrize("S,shift", [(4, 7.0), (50, 7.0), (50, 3.0)])
def test_sigmas(S, shift):
    assert np.allclose(np.array(sampling.sampling_sigmas(S, shift)), osamp.get_sampling_sigmas(S, shift), rtol=0, atol=1e-15)


def test_euler_5b_and_14b():
    g = torch.Generator().manual_seed(0)
    lfz, S = 3, 6
    model_input = torch.randn(6, 7, 4, 5, generator=g)
    noise = torch.randn(6, 7, 4, 5, generator=g)
    sig = sampling.sampling_sigmas(S, 7.0)
    f = fake_field(1)
    lat0 = torch.cat([model_input[:, :-lfz], noise[:, -lfz:]], dim=1)
    want = osamp.euler_5b(f, lat0, model_input, sig, lfz)
    got = sampling.ode_chunk(lambda lat, i: f(lat, i, "cond"), lat0, sig, lfz, sampling.clean_history(model_input[:, :-lfz]))
    assert torch.equal(got, want)
    sig = sampling.sampling_sigmas(S, 3.0)
    lat0 = noise.clone()
    want = osamp.euler_14b(f, lat0, model_input, noise, sig, lfz)
    vel = lambda lat, i: f(lat, i, "uncond") + 5.0 * (f(lat, i, "cond") - f(lat, i, "uncond"))
    got = sampling.ode_chunk(vel, lat0, sig, lfz, sampling.renoised_history(model_input[:, :-lfz], noise[:, :-lfz], sig))
    assert torch.allclose(got, want, atol=1e-6)


@pytest.mark.parametrize("S,cfg,renoise", [(50, True, True), (50, False, False), (12, False, False)])
def test_sde_time_travel(S, cfg, renoise):
    g = torch.Generator().manual_seed(3)
    lfz = 2
    model_input = torch.randn(6, 5, 3, 4, generator=g)
    noise = torch.randn(6, 5, 3, 4, generator=g)
    sig = sampling.sampling_sigmas(S, 3.0)
    f = fake_field(2)
    calls = {"n": 0}

    def vel(lat, i):
        calls["n"] += 1
        c = f(lat, i, "cond")
        if not cfg:
            return c
        u = f(lat, i, "uncond")
        return u + 5.0 * (c - u)

    lat0 = noise.clone() if renoise else torch.cat([model_input[:, :-lfz], noise[:, -lfz:]], dim=1)
    want = osamp.tts(f, lat0, model_input, noise, sig, lfz, Replay(9), sde=True, cfg=cfg, renoise=renoise)
    hist = (sampling.renoised_history(model_input[:, :-lfz], noise[:, :-lfz], sig) if renoise
            else sampling.clean_history(model_input[:, :-lfz]))
    got = sampling.sde_tts_chunk(vel, lat0, sig, lfz, hist, sde=True, generator=torch.Generator().manual_seed(9))
    assert torch.allclose(got, want, atol=1e-5), (got - want).abs().max()
    assert calls["n"] == sampling.tts_forward_count(S)
    if S == 50:
        assert calls["n"] == 74


# ---- pinned to the reference's sampling SCRIPTS: their own loops, cut out of the script text and executed (oracle/ref_scripts.py) ----
from oracle import ref_scripts  # noqa: E402

GOLD = __import__("os").path.join(ROOT, "tests", "golden", "sampler_scripts.pt")


def _script_inputs(fx, dtype):
    c = ref_scripts.script_case(fx["seed_case"], dtype)
    return c, ref_scripts.script_field(fx["seed_field"], c["C"]), c["model_input"], c["noise"], fx["lfz"]


def _replay(seed, dtype):
    g = torch.Generator().manual_seed(seed)
    return lambda shape: torch.randn(shape, generator=g, dtype=dtype)


@pytest.mark.skipif(not ref_scripts.available(), reason="needs the reference tree (build container only)")
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_oracle_sampler_is_bit_identical_to_the_reference_scripts_live(dtype):
    """oracle/sampler.py against sample_tts.py / sample.py / sample_5b.py's OWN loops executed here: same bits, same number of model
    calls in the same order, and the 5B script's per-token timestep vector is zeros on the history tokens, 1000*sigma_i behind them."""
    fx = torch.load(GOLD, weights_only=False)
    c, f, mi, noise, lfz = _script_inputs(fx, dtype)
    sig3 = list(osamp.get_sampling_sigmas(50, 3.0))
    for sde in (True, False):
        ref, draws, calls, span = ref_scripts.run_tts(f, noise.clone(), noise, mi, sig3, lfz, sde=sde, seed=fx["seed_noise"])
        seen = []
        got = osamp.tts(lambda l, i, w: (seen.append((i, w)), f(l, i, w))[1], noise.clone(), mi, noise, sig3, lfz,
                        _replay(fx["seed_noise"], dtype), sde=sde)
        assert torch.equal(got, ref) and seen == calls and len(calls) == 148 and len(draws) == (74 if sde else 0)
    for n in (50, 6):
        sig = list(osamp.get_sampling_sigmas(n, 3.0))
        ref, calls, _ = ref_scripts.run_euler_14b(f, noise.clone(), noise, mi, sig, lfz)
        assert torch.equal(osamp.euler_14b(f, noise.clone(), mi, noise, sig, lfz), ref) and len(calls) == 2 * n
        sig = list(osamp.get_sampling_sigmas(n, 7.0))
        lat0 = torch.cat([mi[:, :-lfz], noise[:, -lfz:]], dim=1)
        n0 = (c["F"] - lfz) * (c["H"] // 2) * (c["W"] // 2)
        seq_len = c["F"] * (c["H"] // 2) * (c["W"] // 2) + 5
        ref, tvecs, calls, _ = ref_scripts.run_euler_5b(f, lat0.clone(), mi, sig, lfz, seq_len)
        assert torch.equal(osamp.euler_5b(f, lat0.clone(), mi, sig, lfz), ref) and len(calls) == n
        for i, t in enumerate(tvecs):
            want = torch.cat([torch.zeros(n0, dtype=torch.float64), torch.ones(seq_len - n0, dtype=torch.float64) * (sig[i] * 1000)])
            assert t.shape == (1, seq_len) and t.dtype == torch.float64 and torch.equal(t[0], want)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_oracle_and_host_loops_match_the_reference_script_fixture(dtype):
    """the same check from the committed fixture (oracle/make_golden_sampler.py ran the scripts' loops in the build container), plus the
    product's host loops (yume_amd/sampling.py) against the scripts' results."""
    fx = torch.load(GOLD, weights_only=False)
    res = fx["cases"][str(dtype).split(".")[-1]]
    c, f, mi, noise, lfz = _script_inputs(fx, dtype)
    # the fixture was computed on the build container's CPU: tanh / einsum may round differently in the last bit on another CPU, so
    # "equal" here is a tolerance a few ulps wide after 74 chained steps (the live test above is bit-exact on one machine)
    tol = dict(rtol=0, atol=1e-10) if dtype == torch.float64 else dict(rtol=0, atol=1e-4)
    same = lambda a, b: torch.allclose(a, b, **tol)
    sig3 = list(osamp.get_sampling_sigmas(50, 3.0))
    vel = lambda lat, i: f(lat, i, "uncond") + 5.0 * (f(lat, i, "cond") - f(lat, i, "uncond"))
    hist3 = sampling.renoised_history(mi[:, :-lfz], noise[:, :-lfz], sig3)
    for key, sde in (("tts_50", True), ("tts_50_ode", False)):
        want = res[key]["latent"]
        assert res[key]["n_calls"] == 148 and res[key]["n_draws"] == (74 if sde else 0)
        assert same(osamp.tts(f, noise.clone(), mi, noise, sig3, lfz, _replay(fx["seed_noise"], dtype), sde=sde), want)
        got = sampling.sde_tts_chunk(vel, noise.clone(), sig3, lfz, hist3, sde=sde, generator=torch.Generator().manual_seed(fx["seed_noise"]))
        assert torch.allclose(got, want, **tol), (got - want).abs().max()
    for n in (50, 6):
        sig = list(osamp.get_sampling_sigmas(n, 3.0))
        want = res[f"euler14b_{n}"]["latent"]
        assert same(osamp.euler_14b(f, noise.clone(), mi, noise, sig, lfz), want)
        got = sampling.ode_chunk(vel, noise.clone(), sig, lfz, sampling.renoised_history(mi[:, :-lfz], noise[:, :-lfz], sig))
        assert torch.allclose(got, want, **tol)
        sig = list(osamp.get_sampling_sigmas(n, 7.0))
        want = res[f"euler5b_{n}"]["latent"]
        lat0 = torch.cat([mi[:, :-lfz], noise[:, -lfz:]], dim=1)
        assert same(osamp.euler_5b(f, lat0.clone(), mi, sig, lfz), want)
        got = sampling.ode_chunk(lambda lat, i: f(lat, i, "cond"), lat0.clone(), sig, lfz, sampling.clean_history(mi[:, :-lfz]))
        assert torch.allclose(got, want, **tol)
        # the script's per-token timestep vectors: what make_velocity_5b builds (0 on history tokens, 1000 * sigma_i on the rest, float64)
        t = res[f"euler5b_{n}"]["t"]
        n0 = (c["F"] - lfz) * (c["H"] // 2) * (c["W"] // 2)
        assert t.dtype == torch.float64 and t.shape == (n, 1, res[f"euler5b_{n}"]["seq_len"])
        for i in range(n):
            assert torch.equal(t[i, 0, :n0], torch.zeros(n0, dtype=torch.float64))
            assert torch.equal(t[i, 0, n0:], torch.ones(t.shape[2] - n0, dtype=torch.float64) * (sig[i] * 1000.0))


# ---- the FramePack chunk loop (sample_5b.py:920-1097): product vs the oracle's restatement, stand-in model and VAE on the CPU ----
class _FakeModel:
    """WanModel.forward's call shape for make_velocity_5b (list in, list out, per-token t), a deterministic field inside."""

    def __init__(self, lfz):
        self.lfz, self.p, self.calls = lfz, torch.nn.Parameter(torch.zeros(1)), []

    def parameters(self):
        return iter([self.p])

    def field(self, lat, sigma, ctx):
        return torch.tanh(lat[:, -self.lfz:] * 0.7 + lat[:, :1].mean() + ctx.mean()) * (0.5 + sigma)

    def __call__(self, xs, t, context, seq_len, latent_frame_zero, flag):
        assert flag and latent_frame_zero == self.lfz and t.shape == (1, seq_len) and t[0, 0] == 0
        self.calls.append((xs[0].shape[1], seq_len))
        return [self.field(xs[0], float(t[0, -1]) / 1000.0, context[0])]


def test_long_video_loop_matches_the_oracle_restatement():
    from yume_amd import framepack
    g = torch.Generator().manual_seed(11)
    lfz, steps, shift, F0, H, W = 3, 4, 7.0, 5, 4, 6
    h0 = torch.randn(6, F0, H, W, generator=g)
    ctxs = [torch.randn(7, 8, generator=g) for _ in range(3)]
    sig = sampling.sampling_sigmas(steps, shift)
    fm = _FakeModel(lfz)

    class _Vae:
        def decode(self, zs):
            return [zs[0] * 2.0 + 1.0]
    hist, vids = sampling.long_video_5b(fm, _Vae(), h0, ctxs, steps, shift, lfz, generator=torch.Generator().manual_seed(12))
    g2 = torch.Generator().manual_seed(12)

    def randn(shape):               # the script draws the padded shape and reads its last lfz frames; the product draws those frames only
        out = torch.zeros(tuple(shape))
        out[:, -lfz:] = torch.randn((shape[0], lfz, shape[2], shape[3]), generator=g2)
        return out
    want, wv = osamp.long_video_5b(lambda lat, i, which, k: fm.field(lat, sig[i], ctxs[k]), lambda z: z * 2.0 + 1.0, h0, 3, sig, lfz, randn)
    assert hist.shape == (6, F0 + 3 * lfz, H, W)
    assert torch.allclose(hist, want, atol=1e-6) and all(torch.allclose(a, b, atol=1e-6) for a, b in zip(vids, wv))
    assert torch.equal(hist[:, :F0], h0)
    assert [c[0] for c in fm.calls] == [F0 + lfz] * steps + [F0 + 2 * lfz] * steps + [F0 + 3 * lfz] * steps
    assert [c[1] for c in fm.calls[::steps]] == [framepack.pack_plan(F0 + (k + 1) * lfz, H, W, lfz).seq_len for k in range(3)]
