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


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("family,F,lfz,packed", [("wan23", 15, 8, True), ("wan23", 3, 8, False), ("wan", 16, 9, True)])
def test_oracle_matches_live_reference_without_qk_norm(family, F, lfz, packed):
    """qk_norm=False (wan23/modules/model.py:175-176, wan/modules/model.py WanSelfAttention / WanI2VCrossAttention: nn.Identity in place of
    WanRMSNorm — no norm_q / norm_k / norm_k_img weights in the state dict): the oracle passes q / k through, like the reference."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from make_golden import build_reference, run_reference, token_count
    cfg = dict(synth.tiny_cfg(family, layers=2), qk_norm=False)
    sd = synth.make_dit_state_dict(cfg, family, seed=61)
    assert not any("norm_q" in k or "norm_k" in k for k in sd)
    ref = build_reference(family, cfg, sd)                       # strict load: the reference has no such parameters either
    inp = synth.make_dit_inputs(cfg, family, F, 10, 12, n_text=9, seed=62)
    L = token_count(family, F, 10, 12, lfz, packed)
    if family == "wan23" and packed:
        t = torch.cat([torch.zeros(5), torch.full((L - 5,), 333.25)]).unsqueeze(0).double()
    else:
        t = torch.tensor([250.0])
    want, _ = run_reference(ref, family, inp, t, L, lfz, packed)
    got = run_oracle(dict(family=family, cfg=cfg, inputs=inp, t=t, seq_len=L, lfz=lfz, packed=packed), sd)
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 2e-5
    # and the switch changes the function (the same weights with the norms in place give another output)
    cfg_n = dict(cfg, qk_norm=True)
    sd_n = synth.make_dit_state_dict(cfg_n, family, seed=61)
    other = run_oracle(dict(family=family, cfg=cfg_n, inputs=inp, t=t, seq_len=L, lfz=lfz, packed=packed), sd_n)
    assert (other - got).abs().max().item() > 1e-3


def _cache_case():
    from yume_amd import framepack
    family, F, H, W, lfz = "wan", 13, 10, 12, 9
    cfg = synth.tiny_cfg(family, layers=3)
    sd = synth.make_dit_state_dict(cfg, family, seed=51)
    a = synth.make_dit_inputs(cfg, family, F, H, W, n_text=9, seed=52)
    b = synth.make_dit_inputs(cfg, family, F, H, W, n_text=9, seed=53)
    L = framepack.pack_plan(F, H, W, lfz, F - 9).seq_len
    return family, cfg, sd, a, b, L, lfz, [2, 0]


def _oracle_cache_run(cfg, sd, a, b, L, lfz, cache_list):
    out1, cache = odit.forward_wan(sd, cfg, a["x"], torch.tensor([700.0]), a["context"], L, a["clip_fea"][0], a["y"], 0.6, lfz,
                                   cache_sample=True, return_cache=True, cache_list=cache_list)
    out2, none = odit.forward_wan(sd, cfg, b["x"], torch.tensor([650.0]), b["context"], L, b["clip_fea"][0], b["y"], 0.6, lfz,
                                  cache_sample=True, cache=cache, return_cache=False, cache_list=cache_list)
    assert none is None
    return out1, cache, out2


def test_oracle_block_residual_cache_matches_golden():
    """a13 (wan/modules/model.py:975-1000): fixture generated from the REAL reference by oracle/make_golden_cache.py."""
    fx = load_golden("dit_wan_cache")
    family, cfg, sd, a, b, L, lfz, cache_list = _cache_case()
    out1, cache, out2 = _oracle_cache_run(cfg, sd, a, b, L, lfz, cache_list)
    assert (out1 - fx["out_record"]).abs().max() <= 2e-5 and (out2 - fx["out_replay"]).abs().max() <= 2e-5
    assert len(cache) == len(fx["cache"]) == 2
    for c, r in zip(cache, fx["cache"]):
        assert c.dtype == torch.bfloat16 and c.shape == r.shape and (c.float() - r.float()).abs().max() <= 2e-2 * r.float().abs().max()


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")
def test_oracle_block_residual_cache_matches_live_reference():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from make_golden_cache import reference_cache_run
    family, cfg, sd, a, b, L, lfz, cache_list = _cache_case()
    r1, rc, r2 = reference_cache_run(cfg, sd, a, b, L, lfz, cache_list)
    out1, cache, out2 = _oracle_cache_run(cfg, sd, a, b, L, lfz, cache_list)
    assert (out1 - r1).abs().max() <= 2e-5 and (out2 - r2).abs().max() <= 2e-5
    assert [tuple(c.shape) for c in cache] == [tuple(c.shape) for c in rc]


def test_oracle_block_matches_the_reference_block_at_full_width():
    """tests/golden/block_bf16_deviation.pt (oracle/make_golden_bf16dev.py): the REAL reference WanAttentionBlock at full 5B width
    (dim 3072, 24 heads, ffn 14336), L = 2048, 77 text tokens, fp32 on CPU. The oracle's block restatement — on the head-by-head
    attention evaluation the full-size tests use (oracle/fullsize.py) — reproduces its rows to fp32 round-off."""
    from oracle import fullsize
    fx = load_golden("block_bf16_deviation")
    case = fullsize.make_block_case(synth.CFG_5B, "wan23", fx["L"], seed=fx["seed"], n_text=fx["n_text"])
    assert abs(float(case["x"].double().sum()) - fx["x_checksum"]) <= 1e-9 * abs(fx["x_checksum"]), "synthetic inputs drifted"
    want, _ = fullsize.run_block_oracle(case)
    err = (want[fx["rows"]] - fx["gold_rows"]).abs().max().item()
    assert err <= 5e-5, err
    # and the fixture's own statement of the reference's bf16 deviation is self-consistent on the stored rows
    x = case["x"][fx["rows"]].double()
    d = (fx["bf16_rows"].double() - fx["gold_rows"].double()).norm() / (fx["gold_rows"].double() - x).norm()
    assert 0.5 * fx["reference_bf16_deviation"]["update_rel_l2"] <= d.item() <= 2.0 * fx["reference_bf16_deviation"]["update_rel_l2"]
