"""TEST INFRASTRUCTURE — generates tests/golden/t5_tiny.pt by running the REAL reference T5Encoder (wan/modules/t5.py,
imported from /root/reference) on seeded synthetic weights / token ids.      python oracle/make_golden_t5.py
Weights are regenerated from the seed by yume_amd.synth.make_t5_state_dict."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from yume_amd import synth  # noqa: E402


def build_reference_encoder(cfg, sd):
    mod = ref_import.ref_t5()
    m = mod.T5Encoder(cfg["vocab"], cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"], cfg["num_layers"],
                      cfg["num_buckets"], shared_pos=cfg["shared_pos"], dropout=0.0).eval().requires_grad_(False)
    m.load_state_dict(sd, strict=True)
    return m


def main():
    assert ref_import.available()
    cfg, seed = synth.tiny_t5_cfg(), 3
    sd = synth.make_t5_state_dict(cfg, seed)
    ref = build_reference_encoder(cfg, sd)
    g = torch.Generator().manual_seed(5)
    L = 96
    ids = torch.randint(1, cfg["vocab"], (3, L), generator=g)
    lens = [96, 37, 70]
    mask = torch.zeros(3, L, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        ids[b, n:] = 0
    with torch.no_grad():
        out = ref(ids, mask)
    fx = dict(cfg=cfg, seed=seed, ids=ids, mask=mask, lens=lens, out=out.float())
    path = os.path.join(ROOT, "tests", "golden", "t5_tiny.pt")
    torch.save(fx, path)
    print("wrote", path, tuple(out.shape), float(out.abs().mean()))


if __name__ == "__main__":
    main()
