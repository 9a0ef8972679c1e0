"""Host logic: the product's FramePack plan / RoPE table (yume_amd/framepack.py) against the oracle's independent
restatement of the reference branches (oracle/dit.py::framepack_plan, rope_grid)."""
import sys

import pytest
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)
from oracle import dit as odit  # noqa: E402
from yume_amd import framepack  # noqa: E402

_LEVEL = {"": 0, "_2x": 1, "_4x": 2, "_8x": 3, "_16x": 4, "2x_f+_16x": 5}


@pytest.mark.parametrize("n_hist", list(range(1, 100)) + [341, 342, 343, 400, 1000, 1366])
def test_plan_matches_oracle(n_hist):
    got = framepack.history_groups(n_hist, 44, 80)
    want = odit.framepack_plan(n_hist)
    assert len(got) == len(want)
    for g, (sl, name) in zip(got, want):
        assert (g.f0, g.f0 + g.nf) == (sl.start, sl.stop), (n_hist, g, sl)
        assert g.level == _LEVEL[name]


def test_plan_limits():
    with pytest.raises(ValueError):
        framepack.history_groups(1367, 44, 80)
    with pytest.raises(ValueError):
        framepack.pack_plan(8, 44, 80, 8)   # no history


def test_sequence_lengths_of_the_long_video_loop():
    # L per chunk for the 5B model at 704x1280
    want = [9460, 11420, 11900, 12065, 12185, 12305, 12425, 12545]
    got = [framepack.pack_plan(13 + 8 * k, 44, 80, 8).seq_len for k in range(8)]
    assert got == want
    assert framepack.pack_plan(13, 68, 120, 9, 13 - 9).seq_len == 23460   # 14B-c0
    assert framepack.pack_plan(17, 68, 120, 9, 17 - 9).seq_len == 27810   # 14B 65-frame clip


@pytest.mark.parametrize("F,H,W,lfz", [(13, 12, 16, 8), (21, 44, 80, 8), (40, 10, 14, 8), (110, 10, 14, 8), (110, 6, 10, 8), (14, 12, 16, 9)])
def test_rope_table_matches_oracle(F, H, W, lfz):
    plan = framepack.pack_plan(F, H, W, lfz)
    cs = framepack.plan_rope(plan, 128)                      # [L, 64, 2] fp32
    tabs = odit.rope_axes(128)
    parts, f_off = [], 0
    for g in plan.groups:
        parts.append(odit.rope_grid(tabs, g.nf, g.hp, g.wp, f_off))
        f_off += g.nf
    want = torch.cat(parts)                                   # complex128 [L, 64]
    assert cs.shape == (plan.seq_len, 64, 2)
    assert torch.allclose(cs[..., 0].double(), want.real, atol=1e-7)
    assert torch.allclose(cs[..., 1].double(), want.imag, atol=1e-7)


def test_14b_branch_selector():
    # wan/modules/model.py:781 selects with f_num - 9 while slicing with latent_frame_zero
    g9 = framepack.history_groups(16 - 9, 10, 12, 16 - 9)
    g8 = framepack.history_groups(16 - 8, 10, 12, 16 - 9)
    assert [x.level for x in g9] == [x.level for x in g8]
    w = odit.framepack_plan(16 - 8, 16 - 9)
    assert [(g.f0, g.f0 + g.nf) for g in g8] == [(s.start, s.stop) for s, _ in w]
