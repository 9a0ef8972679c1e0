"""CLIP ViT-H/14 vision tower: CPU oracle against the golden vectors of the real reference class and
against the class itself; the drop-in's parameter names; the device path against the golden on the GPU."""
import sys

import pytest
import torch

from conftest import ROOT, load_golden

sys.path.insert(0, ROOT)
from oracle import clip as oclip, ref_import  # noqa: E402
from yume_amd import clip, synth  # noqa: E402


def test_oracle_matches_reference_golden():
    fx = load_golden("clip_tiny")
    sd = synth.make_clip_state_dict(fx["cfg"], fx["seed"])
    for flag, key in ((True, "out31"), (False, "out")):
        got = oclip.visual_forward(sd, fx["cfg"], fx["x"], use_31_block=flag)
        assert (got - fx[key]).abs().max() <= 2e-5 * fx[key].abs().max()


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present")
def test_oracle_matches_reference_class_directly():
    cfg = synth.tiny_clip_cfg(dim=160, heads=2, layers=2, image=42)
    sd = synth.make_clip_state_dict(cfg, 2)
    ref = ref_import.ref_clip().VisionTransformer(**cfg).eval()
    ref.load_state_dict(sd, strict=True)
    x = torch.randn(1, 3, 42, 42, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = ref(x, use_31_block=True)
    got = oclip.visual_forward(sd, cfg, x, True)
    assert (got - want).abs().max() <= 2e-5 * want.abs().max()


def test_dropin_has_the_reference_parameter_names():
    cfg = synth.tiny_clip_cfg()
    m = clip.VisionTransformer(**cfg)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == synth.clip_param_shapes(cfg)
    big = synth.clip_param_shapes(synth.CLIP_CFG_VIT_H)
    assert big["pos_embedding"] == (1, 257, 1280) and big["transformer.31.attn.to_qkv.weight"] == (3840, 1280)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, cfg["image_size"], cfg["image_size"]))


@pytest.mark.gpu
def test_clip_visual_matches_reference_golden_on_gpu():
    fx = load_golden("clip_tiny")
    cfg = fx["cfg"]
    with torch.device("cuda"):
        m = clip.VisionTransformer(**cfg)
    m.load_state_dict(synth.make_clip_state_dict(cfg, fx["seed"]), strict=True)
    m = m.cuda().eval()
    for flag, key in ((True, "out31"), (False, "out")):
        got = m(fx["x"].cuda(), use_31_block=flag).float().cpu()
        e = ((got.double() - fx[key].double()).norm() / fx[key].double().norm()).item()
        print(f"clip tiny use_31_block={flag}: rel-L2 {e:.3e}")
        assert got.shape == fx[key].shape and e <= 1e-2
    # the wrapper's preprocessing + call shape (image2video.py:341: clip.visual([img[:, None, :, :]]))
    wrap = clip.CLIPModel(device="cuda", model=m)
    img = torch.rand(3, 1, 70, 90, device="cuda") * 2 - 1
    out = wrap.visual([img])
    assert out.shape == (1, 17, cfg["dim"]) and torch.isfinite(out).all()
