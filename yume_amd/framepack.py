"""FramePack history packing schedule and per-token RoPE table (host side, built once per clip shape).

Mirrors what WanModel.forward does at the top of every call in the reference
(wan23/modules/model.py:588-741, up_fre :933-940; wan/modules/model.py:768-910): older latent frames are
patch-embedded with geometrically larger kernels, the newest `latent_frame_zero` frames at (1,2,2), and each
token carries its own complex RoPE phase whose temporal index keeps counting across the packed groups
while the spatial indices restart at 0 on every (down-sampled) grid.

The reference rebuilds all of this (including three rope_params(1024, .) tables on the CPU and a host->device
copy) on every denoise step; here it is a pure function of (F, H, W, latent_frame_zero), computed once and
cached on the device.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

# level l patch-embeds with a (1, 2<<l, 2<<l) kernel: 0 -> patch_embedding, 1 -> _2x, 2 -> _4x, 3 -> _8x, 4 -> _16x,
# 5 -> patch_embedding_2x_f followed by _16x.
LEVEL_SUFFIX = ["", "_2x", "_4x", "_8x", "_16x"]
_BRANCH_LIMIT = [6, 22, 86, 342, 1366]      # history length admitted by branch 1..5
_TAIL_CUTS = [3, 5, 21, 85, 341]            # frames-from-the-end boundaries of the tail groups (branch >= 2)


@dataclass(frozen=True)
class Group:
    f0: int          # first source frame (index into the full clip)
    nf: int          # number of source frames
    level: int       # see LEVEL_SUFFIX; 5 = composite 2x_f -> 16x
    hp: int          # patch grid height
    wp: int          # patch grid width

    @property
    def ntok(self):
        return self.nf * self.hp * self.wp

    @property
    def ksize(self):
        return 2 << min(self.level, 4)


def _grid(h, w, level):
    """token grid of a level. Level 0 is the reference's UNPADDED stride-2 patch_embedding (wan23/modules/model.py:455): it
    floors, so odd sizes would silently lose a row / column there while the pyramid levels zero-pad (convpadd, :588-741);
    every shipped resolution is even and an odd one is refused rather than guessed."""
    if level == 0 and (h % 2 or w % 2):
        raise ValueError(f"latent H x W = {h} x {w} must be even for the (1, 2, 2) patch embedding")
    if level == 5:
        h, w = -(-h // 4), -(-w // 4)
        level = 4
    k = 2 << level
    return -(-h // k), -(-w // k)


def history_groups(n_hist: int, h: int, w: int, n_sel: Optional[int] = None) -> List[Group]:
    """Groups covering the n_hist history frames, oldest first. `n_sel` selects the branch (the 14B model
    hard-codes `f_num - 9` there, wan/modules/model.py:781) while the slices always use n_hist."""
    sel = n_hist if n_sel is None else n_sel
    branch = next((b for b, lim in enumerate(_BRANCH_LIMIT, 1) if sel <= lim), None)
    if branch is None:
        raise ValueError(f"{sel} history latent frames exceed the deepest FramePack level (1366)")
    cuts = [1] if branch == 1 else _TAIL_CUTS[:branch]
    first_level = 0 if branch <= 3 else 1
    groups = [Group(0, 1, first_level, *_grid(h, w, first_level))]
    # the "middle" run between the first frame and the oldest tail group, at the coarsest level
    far = cuts[-1]
    mid_level = len(cuts)
    if n_hist - (far + 1) <= 0:
        mf0, mnf = n_hist - far, 1          # degenerate: the single frame u1[:, :, -far]
    else:
        mf0, mnf = 1, n_hist - far - 1      # u1[:, :, 1:-far]
    groups.append(Group(mf0, mnf, mid_level, *_grid(h, w, mid_level)))
    # tail groups, oldest (coarsest) first: [-cuts[i] : -cuts[i-1]) at level i
    for i in range(len(cuts) - 1, -1, -1):
        lo = n_hist - cuts[i]
        hi = n_hist - (cuts[i - 1] if i > 0 else 0)
        groups.append(Group(lo, hi - lo, i, *_grid(h, w, i)))
    for g in groups:
        if g.f0 < 0 or g.nf <= 0 or g.f0 + g.nf > n_hist:
            raise ValueError(f"FramePack group {g} does not fit a history of {n_hist} frames")
    return groups


@dataclass
class PackPlan:
    groups: List[Group]            # history groups then the new-frames group (last)
    n_hist_tok: int
    n_new_tok: int
    new_grid: Tuple[int, int, int]  # (F, Hp, Wp) of the frames being denoised

    @property
    def seq_len(self):
        return self.n_hist_tok + self.n_new_tok


def pack_plan(n_frames: int, h: int, w: int, lfz: int, n_sel: Optional[int] = None) -> PackPlan:
    n_hist = n_frames - lfz
    if n_hist <= 0:
        raise ValueError(f"FramePack path needs history: {n_frames} frames, latent_frame_zero={lfz}")
    gs = history_groups(n_hist, h, w, n_sel)
    n_hist_tok = sum(g.ntok for g in gs)
    new = Group(n_hist, lfz, 0, *_grid(h, w, 0))
    return PackPlan(gs + [new], n_hist_tok, new.ntok, (lfz, new.hp, new.wp))


def rope_axis_angles(head_dim: int, max_len: int = 1024, theta: float = 10000.0):
    """fp64 angle tables of the three axes (wan23/modules/model.py:27-35,475-480): widths d-4(d//6), 2(d//6), 2(d//6)."""
    out = []
    for dim in (head_dim - 4 * (head_dim // 6), 2 * (head_dim // 6), 2 * (head_dim // 6)):
        inv = 1.0 / torch.pow(theta, torch.arange(0, dim, 2, dtype=torch.float64) / dim)
        out.append(torch.outer(torch.arange(max_len, dtype=torch.float64), inv))
    return out


def rope_cos_sin(grids: List[Tuple[int, int, int, int]], head_dim: int) -> torch.Tensor:
    """grids: [(f_offset, nf, nh, nw)] in token order. Returns fp32 [L, head_dim/2, 2] = (cos, sin) per token,
    evaluated in fp64 and rounded once."""
    af, ah, aw = rope_axis_angles(head_dim)
    parts = []
    for f_off, nf, nh, nw in grids:
        if f_off + nf > af.shape[0] or nh > ah.shape[0] or nw > aw.shape[0]:
            raise ValueError("RoPE index exceeds the reference's 1024-entry tables")
        ang = torch.cat([
            af[f_off:f_off + nf].view(nf, 1, 1, -1).expand(nf, nh, nw, -1),
            ah[:nh].view(1, nh, 1, -1).expand(nf, nh, nw, -1),
            aw[:nw].view(1, 1, nw, -1).expand(nf, nh, nw, -1)], dim=-1).reshape(nf * nh * nw, -1)
        parts.append(ang)
    ang = torch.cat(parts)
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).float().contiguous()


def plan_rope(plan: PackPlan, head_dim: int) -> torch.Tensor:
    grids, f_off = [], 0
    for g in plan.groups:
        grids.append((f_off, g.nf, g.hp, g.wp))
        f_off += g.nf
    return rope_cos_sin(grids, head_dim)
