"""Drop-in boundary on the GPU: the WanAttentionBlock.forward seam with the reference's arguments, the 14B block-residual
cache (a13) against the oracle extended the same way, and the flash_attention seam at head_dim < 128."""
import math
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)

from oracle import dit as odit  # noqa: E402
from yume_amd import synth  # noqa: E402

DEV = "cuda"


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def _model(family, cfg, sd):
    if family == "wan23":
        from yume_amd.wan23.modules.model import WanModel
    else:
        from yume_amd.wan.modules.model import WanModel
    with torch.device(DEV):
        m = WanModel(**cfg)
    m.load_state_dict(sd, strict=False)
    return m.eval().requires_grad_(False)


@pytest.mark.parametrize("family", ["wan23", "wan"])
@pytest.mark.parametrize("flag", [True, False])
def test_block_forward_seam_matches_oracle(family, flag):
    """reference signature: block(x, e, seq_lens, grid_sizes, freqs, context, context_lens, ...) — wan23/modules/model.py:272-285,
    wan/modules/model.py:444-459. flag / rand_num_img select per-token phases vs the [1024, 64] table + grid."""
    cfg = synth.tiny_cfg(family, layers=2)
    sd = synth.make_dit_state_dict(cfg, family, seed=31, pyramid=())
    m = _model(family, cfg, sd)
    C, N = cfg["dim"], cfg["num_heads"]
    f, h, w = 3, 5, 7
    L, pad = f * h * w, 11
    g = torch.Generator().manual_seed(32)
    x = torch.randn(1, L + pad, C, generator=g)
    ctx = torch.randn(1, (257 if family == "wan" else 0) + 40, C, generator=g)
    tabs = odit.rope_axes(C // N)
    table = torch.cat([t[:1024] for t in tabs], dim=1)                       # the model's self.freqs layout [1024, 64]
    grid_rope = odit.rope_grid(tabs, f, h, w, 0)                             # [L, 64] complex128
    if family == "wan23":
        e = torch.randn(1, L + pad, 6, C, generator=g) * 0.1
        e6 = e[0, :L]
    else:
        e = torch.randn(1, 6, C, generator=g) * 0.1
        e6 = e[0]
    for i in (0, 1):
        want = odit.block_forward(sd, f"blocks.{i}.", x[0, :L], e6, grid_rope, ctx[0], cfg, family)
        blk = m.blocks[i]
        seq, grids = torch.tensor([L]), torch.tensor([[f, h, w]])
        freqs = grid_rope.unsqueeze(1).to(DEV) if flag else table.to(DEV)
        if family == "wan23":
            got = blk(x.to(DEV), e.to(DEV), seq, grids, freqs, ctx.to(DEV), None, flag=flag)
        else:
            got = blk(x.to(DEV), e.to(DEV), seq, grids, freqs, ctx.to(DEV), None, rand_num_img=0.6 if flag else 0.2)
        assert got.shape == x.shape and got.dtype == x.dtype
        assert torch.equal(got[0, L:].cpu(), x[0, L:])                        # padding rows untouched
        assert rel_l2(got[0, :L].cpu(), want) <= 1e-2
    # context_lens restricts the text keys
    if family == "wan23":
        got = m.blocks[0](x.to(DEV), e.to(DEV), torch.tensor([L]), grids, grid_rope.unsqueeze(1).to(DEV), ctx.to(DEV), torch.tensor([17]), flag=True)
        want = odit.block_forward(sd, "blocks.0.", x[0, :L], e6, grid_rope, ctx[0, :17], cfg, family)
        assert rel_l2(got[0, :L].cpu(), want) <= 1e-2


def test_block_residual_cache_matches_oracle():
    """a13 — wan/modules/model.py:975-1000: record bf16 (x_out - x_in) of the listed blocks, replay them on a later call."""
    from yume_amd import framepack
    family = "wan"
    cfg = synth.tiny_cfg(family, layers=4)
    sd = synth.make_dit_state_dict(cfg, family, seed=41)
    m = _model(family, cfg, sd).attach_pyramid()
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV)
    F_, H, W, lfz = 13, 12, 16, 9
    inp = synth.make_dit_inputs(cfg, family, F_, H, W, n_text=20, seed=42)
    inp2 = synth.make_dit_inputs(cfg, family, F_, H, W, n_text=20, seed=43)
    L = framepack.pack_plan(F_, H, W, lfz, F_ - 9).seq_len
    cache_list = [2, 0]                                                       # unsorted on purpose: replay indexes by cache_list order
    t1, t2 = torch.tensor([700.0]), torch.tensor([650.0])

    def dev_call(i, t, **kw):
        return m([i["x"].to(DEV)], t=t.to(DEV), context=[i["context"].to(DEV)], seq_len=L, clip_fea=i["clip_fea"].to(DEV),
                 y=[i["y"].to(DEV)], rand_num_img=0.6, latent_frame_zero=lfz, **kw)

    def ora_call(i, t, **kw):
        return odit.forward_wan(sd, cfg, i["x"], t, i["context"], L, i["clip_fea"][0], i["y"], 0.6, lfz, **kw)

    out_d, cache_d = dev_call(inp, t1, cache_sample=True, return_cache=True, cache_list=cache_list)
    out_o, cache_o = ora_call(inp, t1, cache_sample=True, return_cache=True, cache_list=cache_list)
    assert len(cache_d) == len(cache_o) == 2 and cache_d[0].dtype == torch.bfloat16 and cache_d[0].shape == cache_o[0].shape
    assert rel_l2(out_d.cpu(), out_o) <= 1.5e-2
    for a, b in zip(cache_d, cache_o):
        assert rel_l2(a.float().cpu(), b.float()) <= 3e-2
    # recording does not change the output
    plain, none = dev_call(inp, t1)
    assert none is None and torch.equal(plain, out_d)
    # replay on other inputs: listed blocks are replaced by the stored residuals (cache[cache_list.index(block)])
    rep_d, c2 = dev_call(inp2, t2, cache_sample=True, cache=cache_d, return_cache=False, cache_list=cache_list)
    rep_o, _ = ora_call(inp2, t2, cache_sample=True, cache=cache_o, return_cache=False, cache_list=cache_list)
    assert c2 is None
    assert rel_l2(rep_d.cpu(), rep_o) <= 2e-2
    assert rel_l2(rep_d.cpu(), dev_call(inp2, t2)[0].cpu()) > 1e-3             # and it really is a different computation
    # the reference's IndexError when replaying without a cache
    with pytest.raises(IndexError):
        dev_call(inp2, t2, cache_sample=True, cache=None, return_cache=False, cache_list=cache_list)


def test_flash_attention_seam_head_dim_80():
    """the reference asserts only head_dim <= 256 (wan/modules/attention.py:54); its CLIP tower calls the seam with 80."""
    from yume_amd.attention import flash_attention
    B, Lq, Lk, H, D = 1, 257, 257, 4, 80
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(B, L, H, D, generator=g) for L in (Lq, Lk, Lk))
    out = flash_attention(q.to(DEV), k.to(DEV), v.to(DEV))
    assert out.shape == (B, Lq, H, D) and out.dtype == torch.float32
    qd, kd, vd = (t[0].transpose(0, 1).double() for t in (q, k, v))
    want = (torch.softmax(qd @ kd.transpose(1, 2) / math.sqrt(D), dim=-1) @ vd).transpose(0, 1)
    assert (out[0].cpu().double() - want).abs().max() <= 2e-2 * want.abs().max()
    with pytest.raises(NotImplementedError, match="head_dim 160"):
        flash_attention(torch.zeros(1, 4, 2, 160, device=DEV), torch.zeros(1, 4, 2, 160, device=DEV), torch.zeros(1, 4, 2, 160, device=DEV))
