"""Pins the CPU VAE oracle (oracle/vae.py) against the committed golden vectors generated from the REAL reference,
and against the real reference itself when /root/reference is present."""
import os
import sys

import pytest
import torch

from conftest import ROOT, load_golden

sys.path.insert(0, ROOT)
from oracle import ref_import, vae as ovae  # noqa: E402
from yume_amd import synth  # noqa: E402


@pytest.mark.parametrize("name", ["vae_22", "vae_21"])
def test_vae_oracle_matches_golden(name):
    fx = load_golden(name)
    sd = synth.make_vae_state_dict(fx["cfg"], fx["seed"])
    cs = sum(float(sd[k].double().abs().sum()) for k in sorted(sd))
    assert abs(cs - fx["weight_checksum"]) <= 1e-9 * fx["weight_checksum"]
    dec = ovae.decode(sd, fx["cfg"], fx["z"])
    enc = ovae.encode(sd, fx["cfg"], fx["video"])
    assert dec.shape == fx["dec"].shape and enc.shape == fx["enc"].shape
    assert (dec - fx["dec"]).abs().max() < 2e-5
    assert (enc - fx["enc"]).abs().max() < 2e-5


def test_param_shapes_match_full_size_counts():
    # 704.7 M (2.2: enc 149.6 M, dec 555.0 M), 126.9 M (2.1)
    n22 = sum(torch.Size(s).numel() for s in synth.vae_param_shapes(synth.VAE_CFG_22).values())
    n21 = sum(torch.Size(s).numel() for s in synth.vae_param_shapes(synth.VAE_CFG_21).values())
    assert abs(n22 / 1e6 - 704.7) < 0.1 and abs(n21 / 1e6 - 126.9) < 0.1


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("ver,T", [("2.2", 1), ("2.2", 5), ("2.1", 2), ("2.1", 4)])
def test_vae_oracle_matches_live_reference(ver, T):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from make_golden_vae import build_reference_vae
    cfg = synth.tiny_vae_cfg(ver, dim=16)
    if ver == "2.2":
        cfg["dec_dim"] = 16
    sd = synth.make_vae_state_dict(cfg, 31)
    ref = build_reference_vae(cfg, sd)
    mean, inv = ovae.latent_scale(ver)
    g = torch.Generator().manual_seed(T)
    z = torch.randn(cfg["z_dim"], T, 2, 4, generator=g)
    s = 16 if ver == "2.2" else 8
    video = torch.rand(3, 1 + 4 * (T - 1) + 2, 2 * s, 4 * s, generator=g) * 2 - 1
    with torch.no_grad():
        wd = ref.decode(z.unsqueeze(0), [mean, inv])[0].clamp(-1, 1)
        we = ref.encode(video.unsqueeze(0), [mean, inv])[0]
    assert (ovae.decode(sd, cfg, z) - wd).abs().max() < 2e-5
    assert (ovae.encode(sd, cfg, video) - we).abs().max() < 2e-5
