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
