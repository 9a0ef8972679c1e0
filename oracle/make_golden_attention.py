"""TEST INFRASTRUCTURE — the operator seam against the reference's OWN attention function.

flash_attention() (wan/modules/attention.py:24-130) needs CUDA and the un-vendored flash-attn wheel, but the same file ships
`attention()` (:133-179) with the same signature, which the reference itself uses where flash-attn is missing: q/k/v cast to `dtype`,
torch scaled_dot_product_attention, output [B, Lq, N, D]. That function runs on CPU. This script calls it (fp32 gold: dtype=float32;
and its default bf16) on seeded inputs whose values are bf16-exact and stores inputs + outputs as tests/golden/attention_seam.pt, so the
GPU box can hold yume_amd.attention.flash_attention / attention to it.

    python oracle/make_golden_attention.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

CASES = [  # B, Lq, Lk, N, D   (inputs are regenerated from the seed by the test: only the outputs are stored)
    (1, 257, 257, 4, 80),       # the reference's CLIP tower calls the seam with head_dim 80 (wan/modules/clip.py)
    (2, 150, 90, 3, 128),
    (1, 300, 2100, 2, 128),     # long enough for the one-wave-per-SIMD kernel
    (1, 64, 512, 8, 128),       # cross-attention: 512 context tokens
]
SEED = 2024


def inputs(case, g):
    B, Lq, Lk, N, D = case
    return tuple(torch.randn(B, L, N, D, generator=g).to(torch.bfloat16) for L in (Lq, Lk, Lk))


def ref_attention_module(family="wan"):
    ref_import._stub_diffusers()
    return ref_import._import_pkg(family, ["attention"])["attention"]


def main():
    assert ref_import.available(), "needs the reference tree"
    att = ref_attention_module("wan")
    assert not (att.FLASH_ATTN_2_AVAILABLE or att.FLASH_ATTN_3_AVAILABLE)
    g = torch.Generator().manual_seed(SEED)
    out = []
    for case in CASES:
        q, k, v = (t.float() for t in inputs(case, g))
        gold = att.attention(q, k, v, dtype=torch.float32)
        bf = att.attention(q, k, v)                                  # the reference's GPU dtype flow (bf16 operands and output)
        out.append(dict(shape=case, q_checksum=float(q.double().sum()), gold=gold, bf16=bf))
    path = os.path.join(ROOT, "tests", "golden", "attention_seam.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
