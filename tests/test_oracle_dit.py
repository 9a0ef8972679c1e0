"""Pins the CPU oracle restatement (oracle/dit.py): (1) against the committed golden vectors generated from the REAL
reference (runs anywhere), (2) against the real reference imported from /root/reference (build container only)."""
import os
import sys

import pytest
import torch

from conftest import ROOT, load_golden

sys.path.insert(0, ROOT)
from oracle import dit as odit  # noqa: E402
from oracle import ref_import  # noqa: E402
from yume_amd import synth  # noqa: E402

GOLDEN = ["dit_wan23_packed_f13", "dit_wan23_packed_f21", "dit_wan23_plain_f4", "dit_wan_packed_f13", "dit_wan_plain_f5"]


def run_oracle(fx, sd):
    inp = fx["inputs"]
    if fx["family"] == "wan23":
        return odit.forward_wan23(sd, fx["cfg"], inp["x"], fx["t"], inp["context"], fx["seq_len"], fx["lfz"], fx["packed"])
    return odit.forward_wan(sd, fx["cfg"], inp["x"], fx["t"], inp["context"], fx["seq_len"], inp["clip_fea"][0], inp["y"],
                            0.6 if fx["packed"] else 0.2, fx["lfz"])


def checksum(sd):
    return sum(float(sd[k].double().abs().sum()) for k in sorted(sd))


@pytest.mark.parametrize("name", GOLDEN)
def test_oracle_matches_golden(name):
    fx = load_golden(name)
    sd = synth.make_dit_state_dict(fx["cfg"], fx["family"], fx["seed"])
    assert abs(checksum(sd) - fx["weight_checksum"]) <= 1e-9 * fx["weight_checksum"], "synthetic weights drifted"
    out = run_oracle(fx, sd)
    assert out.shape == fx["out"].shape
    # same torch build -> the restatement reproduces the reference to fp32 round-off
    err = (out - fx["out"]).abs().max().item()
    assert err <= 2e-5, f"{name}: max abs err {err}"


def test_sigmas_match_reference_formula():
    # fastvideo/sample/sample_5b.py:502-506 with numpy
    import numpy as np
    for steps, shift in ((50, 7.0), (4, 7.0), (50, 3.0)):
        s = np.linspace(1, 0, steps + 1)[:steps]
        s = shift * s / (1 + (shift - 1) * s)
        assert np.allclose(np.array(synth.sampling_sigmas(steps, shift)), s, rtol=0, atol=1e-15)


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("family,F,lfz,packed", [("wan23", 15, 8, True), ("wan23", 32, 8, True), ("wan23", 3, 8, False),
                                                 ("wan", 16, 9, True), ("wan", 12, 8, True), ("wan", 3, 9, False),
                                                 ("wan23", 40, 8, True), ("wan23", 110, 8, True), ("wan23", 360, 8, True),
                                                 ("wan", 100, 9, True)])
def test_oracle_matches_live_reference(family, F, lfz, packed):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from make_golden import build_reference, run_reference, token_count
    cfg = synth.tiny_cfg(family, layers=1)
    sd = synth.make_dit_state_dict(cfg, family, seed=21)
    ref = build_reference(family, cfg, sd)
    inp = synth.make_dit_inputs(cfg, family, F, 10, 12, n_text=9, seed=22)
    if family == "wan" and packed and lfz != 9:
        # branch by F-9, split at lfz (sample_tts.py passes 8)
        from yume_amd import framepack
        L = framepack.pack_plan(F, 10, 12, lfz, F - 9).seq_len
    else:
        L = token_count(family, F, 10, 12, lfz, packed)
    if family == "wan23" and packed:
        t = torch.cat([torch.zeros(5), torch.full((L - 5,), 333.25)]).unsqueeze(0).double()
    else:
        t = torch.tensor([250.0])
    want, _ = run_reference(ref, family, inp, t, L, lfz, packed)
    fx = dict(family=family, cfg=cfg, inputs=inp, t=t, seq_len=L, lfz=lfz, packed=packed)
    got = run_oracle(fx, sd)
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 2e-5
