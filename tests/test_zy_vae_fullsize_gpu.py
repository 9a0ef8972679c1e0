"""Row N2 (b), (c): the VAEs at the sizes bench.py quotes, against the device gold (oracle/devgold.py: oracle/vae.py executed on the GPU
in fp32 with its convolutions as tap-sum fp32 matmuls).

  * Wan2.2 (the 5B pipeline's VAE, wan23/modules/vae2_2.py:797-860): a WHOLE chunk decode [48,8,44,80] -> [3,29,704,1280] (485 TFLOP —
    the `vae_decode` line of bench.py), the 17-frame conditioning-clip encode of the long-video loop and a 33-frame encode, all at 704x1280;
  * Wan2.1 (the 14B pipeline's VAE, wan/modules/vae.py:516-568) at production width: decode [16,13,68,120] -> [3,49,544,960] (218.6
    TFLOP) and the 49-frame encode (130 TFLOP).

The device gold is proven first, per VAE, where the CPU oracle is affordable: the full-resolution first-latent decode of each VAE and a
5-frame full-resolution Wan2.2 encode — device gold vs oracle/vae.py on the host cores, asserted <= 1e-5 rel-L2 (printed).

Stated tolerance: rel-L2 <= 3e-2 for every decode / encode, every decoded frame <= 4e-2 (the reference's own bf16-autocast
deviation on VAE decodes is 1.5e-2)."""
import sys
import time

import pytest
import torch

from conftest import ROOT, start_step_jobs

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)

from oracle import devgold  # noqa: E402
from oracle import vae as ovae  # noqa: E402
from yume_amd import synth  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _oracle_jobs():
    """this module sorts directly in front of tests/test_zz_full_step_gpu.py: the two whole-step CPU oracle jobs that module collects
    (4-5 minutes of 2 x 32 host threads) start here, so that they run under this module's GPU work as well."""
    start_step_jobs()
    yield


def _vae(version, seed):
    cfg = synth.VAE_CFG_22 if version == "2.2" else synth.VAE_CFG_21
    sd = synth.make_vae_state_dict(cfg, seed=seed)
    if version == "2.2":
        from yume_amd.wan23.modules.vae2_2 import Wan2_2_VAE, WanVAE_
        m = WanVAE_(dim=cfg["dim"], dec_dim=cfg["dec_dim"], z_dim=cfg["z_dim"], temperal_downsample=cfg["temperal_downsample"])
        m.load_state_dict(sd, strict=True)
        return cfg, sd, Wan2_2_VAE(z_dim=cfg["z_dim"], device=DEV, model=m)
    from yume_amd.wan.modules.vae import WanVAE, WanVAE_
    m = WanVAE_(dim=cfg["dim"], z_dim=cfg["z_dim"], temperal_downsample=cfg["temperal_downsample"])
    m.load_state_dict(sd, strict=True)
    return cfg, sd, WanVAE(device=DEV, model=m)


def _frames(got, want):
    d = got.double() - want.double()
    return [(d[:, t].norm() / want[:, t].double().norm().clamp_min(1e-30)).item() for t in range(want.shape[1])]


@pytest.mark.parametrize("version,zshape", [("2.2", (48, 1, 44, 40)), ("2.1", (16, 1, 68, 120))])
def test_device_gold_reproduces_the_cpu_oracle_first_latent_decode(version, zshape):
    """the proof of the device gold for each decoder: one latent at production height (Wan2.2: 704 x 640 frames — half the width of r5's case, which
    took 64 s of CPU oracle time in a suite that has to stay under 700 s; the whole-chunk tests below run the full 704 x 1280 against the gold)."""
    cfg, sd, vae = _vae(version, 31)
    z = torch.randn(*zshape, generator=torch.Generator().manual_seed(32))
    torch.set_num_threads(min(32, torch.get_num_threads()))
    t0 = time.time()
    want = ovae.decode(sd, cfg, z)
    cpu_s = time.time() - t0
    t0 = time.time()
    gold = devgold.vae_decode(version, z, 31, DEV)
    r = devgold.rel_l2(gold, want)
    print(f"Wan{version} first-latent decode {tuple(want.shape)}: device gold vs CPU oracle rel-L2 {r:.3e} max-abs {(gold - want).abs().max():.3e}; "
          f"CPU {cpu_s:.1f} s, device gold {time.time() - t0:.1f} s")
    assert gold.shape == want.shape and r <= 1e-5
    # the product path against the CPU oracle itself at this size (the r2 full-resolution first-latent test, 20.6 TFLOP for Wan2.2)
    got = vae.decode([z.to(DEV)])[0].cpu()
    rd = devgold.rel_l2(got, want)
    print(f"Wan{version} first-latent decode: device (bf16) vs CPU oracle rel-L2 {rd:.3e} max-abs {(got - want).abs().max():.3e}")
    assert got.shape == want.shape and torch.isfinite(got).all() and got.abs().max() <= 1.0
    assert rd <= 3e-2


def test_device_gold_reproduces_the_cpu_oracle_five_frame_encode():
    """... and for the encoder path (first frame + one cached 4-frame chunk: strided temporal convs, AvgDown3D) at 352 x 640."""
    cfg, sd, _ = _vae("2.2", 33)
    video = torch.rand(3, 5, 352, 640, generator=torch.Generator().manual_seed(34)) * 2 - 1
    torch.set_num_threads(min(32, torch.get_num_threads()))
    t0 = time.time()
    want = ovae.encode(sd, cfg, video)
    cpu_s = time.time() - t0
    gold = devgold.vae_encode("2.2", video, 33, DEV)
    r = devgold.rel_l2(gold, want)
    print(f"Wan2.2 5-frame encode {tuple(want.shape)}: device gold vs CPU oracle rel-L2 {r:.3e}; CPU {cpu_s:.1f} s")
    assert gold.shape == want.shape and r <= 1e-5


def test_whole_wan22_chunk_decode_vs_device_gold():
    """Row N2(b): the benchmarked chunk — 8 latents [48,8,44,80] -> 29 frames of 704 x 1280."""
    cfg, sd, vae = _vae("2.2", 41)
    z = torch.randn(48, 8, 44, 80, generator=torch.Generator().manual_seed(42))
    got = vae.decode([z.to(DEV)])[0].cpu()
    t0 = time.time()
    gold = devgold.vae_decode("2.2", z, 41, DEV)
    gs = time.time() - t0
    assert got.shape == gold.shape == (3, 29, 704, 1280)
    r, pf = devgold.rel_l2(got, gold), _frames(got, gold)
    print(f"whole Wan2.2 chunk decode [48,8,44,80] -> [3,29,704,1280]: rel-L2 {r:.3e} max-abs {(got - gold).abs().max():.3e}; per frame max "
          f"{max(pf):.3e} (frame {pf.index(max(pf))}); device gold {gs:.1f} s")
    assert torch.isfinite(got).all() and got.abs().max() <= 1.0
    assert r <= 3e-2 and max(pf) <= 4e-2


@pytest.mark.parametrize("frames", [17, 33])
def test_wan22_full_resolution_encode_vs_device_gold(frames):
    """the long-video loop's 17-frame conditioning clip (fastvideo/sample/sample_5b.py:920-1097) and a 33-frame clip, 704 x 1280."""
    cfg, sd, vae = _vae("2.2", 43)
    video = torch.rand(3, frames, 704, 1280, generator=torch.Generator().manual_seed(44 + frames)) * 2 - 1
    got = vae.encode([video.to(DEV)])[0].cpu()
    t0 = time.time()
    gold = devgold.vae_encode("2.2", video, 43, DEV)
    gs = time.time() - t0
    assert got.shape == gold.shape == (48, 1 + (frames - 1) // 4, 44, 80)
    r = devgold.rel_l2(got, gold)
    print(f"Wan2.2 {frames}-frame 704x1280 encode -> {tuple(gold.shape)}: rel-L2 {r:.3e} max-abs {(got - gold).abs().max():.3e}; device gold {gs:.1f} s")
    assert torch.isfinite(got).all() and r <= 3e-2


def test_wan21_production_decode_and_encode_vs_device_gold():
    """Row N2(c): the 14B pipeline's VAE at production width — decode [16,13,68,120] -> [3,49,544,960], encode of 49 frames 544 x 960."""
    cfg, sd, vae = _vae("2.1", 51)
    g = torch.Generator().manual_seed(52)
    z = torch.randn(16, 13, 68, 120, generator=g)
    got = vae.decode([z.to(DEV)])[0].cpu()
    t0 = time.time()
    gold = devgold.vae_decode("2.1", z, 51, DEV)
    gs = time.time() - t0
    assert got.shape == gold.shape == (3, 49, 544, 960)
    r, pf = devgold.rel_l2(got, gold), _frames(got, gold)
    print(f"Wan2.1 decode [16,13,68,120] -> [3,49,544,960]: rel-L2 {r:.3e} max-abs {(got - gold).abs().max():.3e}; per frame max {max(pf):.3e}; "
          f"device gold {gs:.1f} s")
    assert torch.isfinite(got).all() and got.abs().max() <= 1.0
    assert r <= 3e-2 and max(pf) <= 4e-2
    video = torch.rand(3, 49, 544, 960, generator=g) * 2 - 1
    got = vae.encode([video.to(DEV)])[0].cpu()
    t0 = time.time()
    gold = devgold.vae_encode("2.1", video, 51, DEV)
    gs = time.time() - t0
    assert got.shape == gold.shape == (16, 13, 68, 120)
    r = devgold.rel_l2(got, gold)
    print(f"Wan2.1 49-frame 544x960 encode -> [16,13,68,120]: rel-L2 {r:.3e} max-abs {(got - gold).abs().max():.3e}; device gold {gs:.1f} s")
    assert torch.isfinite(got).all() and r <= 3e-2
