"""oracle/devgold.py on the build container (no GPU): the pieces the device gold substitutes into the oracle files — the tap-sum
convolution and the head-by-head fp32 attention — against torch's own, and the complete device-gold code path (oracle.dit / oracle.vae
resolved through it, inputs / weights created for a device) forced onto the host for tiny models: it has to reproduce the plain oracle
to fp32 rounding. On the GPU box the same path runs on `cuda` (tests/test_z[yz]_*_gpu.py) and is proven there against the CPU oracle at
full size before it is used where the CPU cannot go."""
import sys

import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT

sys.path.insert(0, ROOT)
from oracle import devgold  # noqa: E402
from oracle import dit as odit  # noqa: E402
from oracle import step_job  # noqa: E402
from oracle import vae as ovae  # noqa: E402
from yume_amd import synth  # noqa: E402


@pytest.mark.parametrize("shape,wshape,stride,pad", [
    ((1, 5, 4, 9, 11), (7, 5, 3, 3, 3), 1, 0),               # the causal 3x3x3 (padded by the caller)
    ((2, 5, 9, 11), (7, 5, 3, 3), 2, 0),                     # downsample: stride 2 on a (0,1,0,1)-padded frame
    ((3, 5, 9, 11), (7, 5, 3, 3), 1, 1),                     # upsample conv: padding=1
    ((1, 6, 3, 8, 12), (4, 6, 1, 4, 4), (1, 4, 4), 0),       # patch embedding: kernel = stride
    ((1, 4, 5, 6, 6), (8, 4, 3, 1, 1), (2, 1, 1), 0),        # strided time_conv
    ((1, 4, 2, 6, 6), (8, 4, 1, 1, 1), 1, 0),                # 1x1x1
])
def test_conv_taps_equals_torch_convolution(shape, wshape, stride, pad):
    g = torch.Generator().manual_seed(len(shape) * 100 + wshape[0])
    x, w, b = torch.randn(shape, generator=g), torch.randn(wshape, generator=g), torch.randn(wshape[0], generator=g)
    want = (F.conv3d if len(wshape) == 5 else F.conv2d)(x, w, b, stride=stride, padding=pad)
    got = devgold.conv_taps(x, w, b, stride, pad)
    assert got.shape == want.shape and got.is_contiguous()
    assert devgold.rel_l2(got, want) <= 2e-6


def test_attention_dev_equals_the_oracle_attention():
    g = torch.Generator().manual_seed(3)
    q, k, v = torch.randn(300, 3, 16, generator=g), torch.randn(77, 3, 16, generator=g), torch.randn(77, 3, 16, generator=g)
    assert devgold.rel_l2(devgold.attention_dev(q, k, v, q_block=128), odit.attention(q, k, v)) <= 2e-6


@pytest.mark.parametrize("version", ["2.2", "2.1"])
def test_vae_through_the_device_gold_path_equals_the_plain_oracle(version):
    cfg = synth.tiny_vae_cfg(version, dim=16)
    sd = synth.make_vae_state_dict(cfg, seed=4)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(cfg["z_dim"], 3, 4, 6, generator=g)
    want = ovae.decode(sd, cfg, z)
    got = devgold.vae_decode(version, z, 4, dev="cpu", cfg=cfg, force=True)
    assert got.shape == want.shape and devgold.rel_l2(got, want) <= 1e-5
    s = 8 * cfg["patch"]
    video = torch.rand(3, 9, 4 * s, 6 * s, generator=g) * 2 - 1
    want = ovae.encode(sd, cfg, video)
    got = devgold.vae_encode(version, video, 4, dev="cpu", cfg=cfg, force=True)
    assert got.shape == want.shape and devgold.rel_l2(got, want) <= 1e-5
    assert ovae.F is F and odit.F is F                      # the substitution ends with the context


@pytest.mark.parametrize("name", ["tiny5b", "tiny14b"])
def test_dit_step_through_the_device_gold_path_equals_the_plain_oracle(name):
    want, _, _ = step_job.oracle_forward(name, "cond", threads=2)
    got, secs, gen = step_job.oracle_forward(name, "cond", threads=2, device="cpu")     # device-gold code path, forced onto the host
    assert got.shape == want.shape and secs > 0
    assert devgold.rel_l2(got, want) <= 1e-5
    assert odit.attention.__name__ == "attention"
