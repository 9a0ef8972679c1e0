"""TEST INFRASTRUCTURE — generates tests/golden/clip_tiny.pt by running the REAL reference VisionTransformer
(wan/modules/clip.py, imported from /root/reference) on seeded synthetic weights / images.   python oracle/make_golden_clip.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from yume_amd import synth  # noqa: E402


def build_reference(cfg, sd):
    mod = ref_import.ref_clip()
    m = mod.VisionTransformer(**{k: v for k, v in cfg.items()}).eval().requires_grad_(False)
    m.load_state_dict(sd, strict=True)
    return m


def main():
    assert ref_import.available()
    cfg, seed = synth.tiny_clip_cfg(), 4
    sd = synth.make_clip_state_dict(cfg, seed)
    ref = build_reference(cfg, sd)
    x = torch.randn(2, 3, cfg["image_size"], cfg["image_size"], generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        out31 = ref(x, use_31_block=True)
        out = ref(x, use_31_block=False)
    path = os.path.join(ROOT, "tests", "golden", "clip_tiny.pt")
    torch.save(dict(cfg=cfg, seed=seed, x=x, out31=out31.float(), out=out.float()), path)
    print("wrote", path, tuple(out31.shape), float(out31.abs().mean()))


if __name__ == "__main__":
    main()
