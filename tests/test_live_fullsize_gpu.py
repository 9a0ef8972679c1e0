"""Full-size LIVE parity (no identity blocks, oracle-anchored) for BASELINE.json configs[1] and [2] and the VAE:

  * one real full-width 5B block at L = 9460 and one 14B block at L = 27810 through DiTEngine._blocks (the kernels, tile
    shapes, query / key splits and scratch buffers of the product path at those sizes) against oracle/dit.py::block_forward
    on identical inputs (oracle/fullsize.py; CPU leg 20-90 s on the GPU box's host cores);
  * a full-resolution Wan2.2 first-latent decode ([48,1,44,80] -> [3,1,704,1280], 20.6 TFLOP) against oracle/vae.py.

Stated tolerances: block update rel-L2 <= 1e-2 and max-abs <= 5e-2 x the output rms scale; VAE decode rel-L2
<= 3e-2 — the same bars as the small-size tests, where the reference's own bf16-autocast deviation is 3.8e-3 / 1.5e-2."""
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)

from oracle import fullsize  # noqa: E402
from oracle import vae as ovae  # noqa: E402
from yume_amd import synth  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("family,L", [("wan23", 9460), ("wan", 27810)])
def test_live_block_at_full_sequence_length(family, L):
    cfg = synth.CFG_5B if family == "wan23" else synth.CFG_14B
    case = fullsize.make_block_case(cfg, family, L, seed=5)
    model = fullsize.build_block_model(case, DEV)
    got = fullsize.run_block_device(case, model, DEV)
    want, secs = fullsize.run_block_oracle(case)
    p = fullsize.parity(got, want)
    upd = fullsize.parity(got - case["x"], want - case["x"])        # the block's own update, without the residual it is added to
    print(f"live block {family} L={L}: rel-L2 {p['rel_l2']:.3e} max-abs {p['max_abs']:.3e} (rms {p['ref_rms']:.3f}); "
          f"update rel-L2 {upd['rel_l2']:.3e}; CPU oracle {secs:.1f} s")
    assert torch.isfinite(got).all()
    assert p["rel_l2"] <= 1e-2 and p["max_abs"] <= 5e-2 * max(1.0, p["ref_rms"])
    assert upd["rel_l2"] <= 1.5e-2
    # the seam with the reference's own arguments gives the same bits as the engine call
    blk = model.blocks[0]
    x = case["x"].to(DEV).unsqueeze(0)
    seq = torch.tensor([L])
    if family == "wan23":
        y = blk(x, case["e6"].to(DEV).unsqueeze(0), seq, None, case["rope"].to(DEV).unsqueeze(1), case["ctx"].to(DEV).unsqueeze(0), None, flag=True)
    else:
        y = blk(x, case["e6"].to(DEV).unsqueeze(0), seq, None, case["rope"].to(DEV).unsqueeze(1), case["ctx"].to(DEV).unsqueeze(0), None,
                rand_num_img=0.6)
    assert torch.equal(y[0].cpu(), got)


def test_full_resolution_first_latent_decode_vs_oracle():
    cfg = synth.VAE_CFG_22
    sd = synth.make_vae_state_dict(cfg, seed=11)
    from yume_amd.wan23.modules.vae2_2 import Wan2_2_VAE, WanVAE_
    m = WanVAE_(dim=cfg["dim"], dec_dim=cfg["dec_dim"], z_dim=cfg["z_dim"], temperal_downsample=cfg["temperal_downsample"])
    m.load_state_dict(sd, strict=True)
    vae = Wan2_2_VAE(z_dim=cfg["z_dim"], device=DEV, model=m)
    g = torch.Generator().manual_seed(12)
    z = torch.randn(48, 1, 44, 80, generator=g)
    got = vae.decode([z.to(DEV)])[0].cpu()
    want = ovae.decode(sd, cfg, z)
    assert got.shape == want.shape == (3, 1, 704, 1280)
    d = (got.double() - want.double())
    rel = (d.norm() / want.double().norm()).item()
    print(f"full-resolution first-latent decode: rel-L2 {rel:.3e} max-abs {d.abs().max():.3e}")
    assert torch.isfinite(got).all() and got.abs().max() <= 1.0
    assert rel <= 3e-2
