"""The operator seam (wan/modules/attention.py) against the reference's OWN functions.

flash_attention() needs CUDA + the un-vendored flash-attn wheel; the same file's attention() (:133-179) — same signature, what the
reference itself runs where flash-attn is missing — executes on CPU. It pins (i) the exact-softmax stand-in the oracle's DiT pin uses
(oracle/ref_import.py::sdpa_standin) and (ii), through tests/golden/attention_seam.pt (oracle/make_golden_attention.py), the product's
yume_amd.attention.flash_attention / attention on the GPU."""
import math
import os
import sys

import pytest
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)
from oracle import make_golden_attention as mga  # noqa: E402
from oracle import ref_import  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "attention_seam.pt")


def _cases():
    fx = torch.load(GOLD, weights_only=False)
    g = torch.Generator().manual_seed(mga.SEED)
    for c in fx:
        q, k, v = mga.inputs(c["shape"], g)
        assert abs(float(q.float().double().sum()) - c["q_checksum"]) < 1e-9, "the seeded inputs differ from the ones the fixture was made from"
        yield c, q, k, v


@pytest.mark.skipif(not ref_import.available(), reason="needs the reference tree (build container only)")
@pytest.mark.parametrize("family", ["wan", "wan23"])
def test_oracle_attention_standin_matches_the_references_own_attention(family):
    att = mga.ref_attention_module(family)
    assert not (att.FLASH_ATTN_2_AVAILABLE or att.FLASH_ATTN_3_AVAILABLE)      # so attention() takes its own path, not flash-attn's
    g = torch.Generator().manual_seed(1)
    for (B, Lq, Lk, N, D) in [(1, 33, 47, 2, 128), (2, 64, 64, 3, 80), (1, 5, 300, 1, 64)]:
        q, k, v = (torch.randn(B, L, N, D, generator=g) for L in (Lq, Lk, Lk))
        want = att.attention(q, k, v, dtype=torch.float32)
        got = ref_import.sdpa_standin(q, k, v)
        assert got.shape == want.shape and got.dtype == want.dtype
        assert (got - want).abs().max() <= 2e-6
        # the reference's bf16 flow deviates from the fp32 gold by what the seam tests allow the device kernel
        assert (att.attention(q, k, v).float() - want).abs().max() <= 2e-2 * want.abs().max()


def test_fixture_is_the_exact_softmax():
    """anywhere (no reference tree needed): the stored outputs of the reference's attention() are the exact softmax of the seeded inputs"""
    for c, q, k, v in _cases():
        got = ref_import.sdpa_standin(q.float(), k.float(), v.float())
        assert (got - c["gold"]).abs().max() <= 2e-6
        assert c["bf16"].dtype == torch.bfloat16 and c["bf16"].shape == c["gold"].shape


@pytest.mark.gpu
def test_device_seam_matches_the_references_own_attention():
    from yume_amd.attention import attention, flash_attention
    for c, q, k, v in _cases():
        D = c["shape"][-1]
        for fn in (flash_attention, attention):
            out = fn(q.cuda(), k.cuda(), v.cuda())
            assert out.dtype == q.dtype and out.shape == c["gold"].shape          # flash_attention returns q.dtype (attention.py:130)
            err = (out.float().cpu() - c["gold"]).abs().max().item()
            ref_err = (c["bf16"].float() - c["gold"]).abs().max().item()          # the reference's own bf16 deviation on these inputs
            assert err <= 2e-2 * c["gold"].abs().max().item(), (c["shape"], err)
            assert err <= 4 * ref_err + 2e-3, (c["shape"], err, ref_err)
        out32 = flash_attention(q.float().cuda(), k.float().cuda(), v.float().cuda(), softmax_scale=1.0 / math.sqrt(D))
        assert out32.dtype == torch.float32
        assert (out32.cpu() - c["gold"]).abs().max() <= 2e-2 * c["gold"].abs().max()
