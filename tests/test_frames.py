"""Post-decode frame conversion: oracle known answers on CPU, HIP kernel bit-exact on the GPU."""
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)
from oracle import frames as oframes  # noqa: E402


def test_oracle_known_answers():
    # x -> round_half_even(clamp(x/2 + 1/2, 0, 1) * 255)
    x = torch.tensor([-1.0, 1.0, 0.0, -2.0, 3.0, 1 / 255, -1 / 255, 0.5, -0.5, 2 / 255 - 1, 1 - 2 / 255, 4 / 255 - 1])
    want = [0, 255, 128, 0, 255, 128, 127, 191, 64, 1, 254, 2]
    # 0.0 -> 127.5 -> 128 (even); 0.5 -> 191.25 -> 191; -0.5 -> 63.75 -> 64; 1/255 -> 128.0
    got = oframes.frames_u8(x.view(1, 1, 1, -1))
    assert got.shape == (1, 1, 12, 1) and got.dtype == np.uint8
    assert got.reshape(-1).tolist() == want
    # half-way cases land on the even neighbour: (k + 0.5)/255 for even k rounds down, for odd k rounds up
    k = torch.arange(0, 255, dtype=torch.float64)
    xs = (((k + 0.5) / 255) * 2 - 1).float()
    got = oframes.frames_u8(xs.view(1, 1, 1, -1)).reshape(-1).astype(int)
    lo, hi = k.numpy().astype(int), k.numpy().astype(int) + 1
    assert ((got == lo) | (got == hi)).all()


def test_oracle_layout():
    v = torch.linspace(-1, 1, 3 * 2 * 4 * 8).view(3, 2, 4, 8)
    got = oframes.frames_u8(v)
    assert got.shape == (2, 4, 8, 3)
    assert got[1, 2, 3, 0] == oframes.frames_u8(v[0:1, 1:2, 2:3, 3:4]).item()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 1, 4, 8), (3, 5, 64, 96), (1, 2, 6, 10), (4, 3, 16, 12), (3, 29, 176, 320)])
def test_frames_u8_bit_exact(shape):
    from yume_amd import video
    g = torch.Generator().manual_seed(7)
    v = torch.randn(shape, generator=g) * 0.8
    flat = v.view(-1)
    n = min(flat.numel(), 510)
    k = torch.arange(n, dtype=torch.float64) / 2                   # every k/2 / 255: all the exact half-way points
    flat[:n] = ((k / 255) * 2 - 1).float()
    got = video.frames_u8(v.cuda())
    assert got.dtype == torch.uint8 and tuple(got.shape) == (shape[1], shape[2], shape[3], shape[0])
    assert np.array_equal(got.cpu().numpy(), oframes.frames_u8(v))


@pytest.mark.gpu
def test_video_processor_matches_reference_call_shape():
    from yume_amd.video import VideoProcessor
    v = (torch.rand(3, 5, 16, 24) * 2 - 1).cuda()
    vp = VideoProcessor(vae_scale_factor=8)
    pil = vp.postprocess_video(v.unsqueeze(0), output_type="pil")          # sample_5b.py:498-499
    assert len(pil) == 1 and len(pil[0]) == 5 and pil[0][0].size == (24, 16) and pil[0][0].mode == "RGB"
    assert np.array_equal(np.asarray(pil[0][3]), oframes.frames_u8(v.cpu())[3])
    u8 = vp.postprocess_video(v.unsqueeze(0), output_type="uint8")
    assert tuple(u8.shape) == (1, 5, 16, 24, 3)
    arr = vp.postprocess_video(v.unsqueeze(0), output_type="np")
    assert arr.shape == (1, 5, 16, 24, 3) and arr.min() >= 0 and arr.max() <= 1
    with pytest.raises(RuntimeError):
        from yume_amd import video
        video.frames_u8(v.cpu())


def test_tiled_decode_overlap_matches_reference_outputs():
    """golden outputs were produced by the reference's own function (oracle/make_golden_tiled.py) around the same stand-in VAE."""
    import os
    from oracle.make_golden_tiled import FakeVae
    from yume_amd.video import tiled_decode_overlap, _tile_spans
    cases = torch.load(os.path.join(ROOT, "tests", "golden", "tiled_decode.pt"))
    assert len(cases) == 5
    for cs in cases:
        got = tiled_decode_overlap(FakeVae(), cs["z"], n_tiles=cs["n_tiles"], image_overlap_size=cs["image_overlap_size"],
                                   latent_frame_zero=cs["latent_frame_zero"])
        assert got.shape == cs["out"].shape
        torch.testing.assert_close(got, cs["out"], rtol=1e-6, atol=1e-6)
    # 5B 720P call shape (webapp_single_gpu.py:830): 80 latent columns, 5 bands of 16 (+2 towards each neighbour)
    assert _tile_spans(80, 5, 2) == [(0, 18), (14, 34), (30, 50), (46, 66), (62, 80)]
    assert _tile_spans(23, 5, 2) == [(0, 7), (3, 12), (8, 17), (13, 21), (17, 23)]


# ---- the web app's own conversion (webapp_single_gpu.py:117-121): pinned by executing the reference function itself -------------------
import os  # noqa: E402

from oracle import ref_scripts  # noqa: E402

WEBAPP_GOLD = os.path.join(ROOT, "tests", "golden", "frames_webapp.pt")


def _webapp_fixture():
    fx = torch.load(WEBAPP_GOLD, weights_only=False)
    v = ref_scripts.webapp_case()
    assert tuple(v.shape) == fx["shape"] and abs(float(v.double().sum()) - fx["checksum"]) < 1e-9
    return v, fx["frames"].numpy()


@pytest.mark.skipif(not ref_scripts.available(), reason="needs the reference tree (build container only)")
def test_webapp_conversion_restated_equals_the_reference_function_live():
    v = ref_scripts.webapp_case()
    want = ref_scripts.run_webapp_postprocess(v.clone())
    assert want.dtype == np.uint8 and want.shape == (v.shape[1], v.shape[2], v.shape[3], v.shape[0])
    assert np.array_equal(oframes.frames_u8_webapp(v), want)
    # it is NOT the diffusers conversion: truncation vs round-half-even differ on most samples
    assert (oframes.frames_u8(v) != want).mean() > 0.3


def test_webapp_conversion_fixture():
    v, want = _webapp_fixture()
    assert np.array_equal(oframes.frames_u8_webapp(v), want)


@pytest.mark.gpu
def test_frames_u8_trunc_bit_exact_vs_the_webapp_function():
    from yume_amd import video
    v, want = _webapp_fixture()
    got = video.frames_u8(v.cuda(), truncate=True)
    assert got.dtype == torch.uint8 and np.array_equal(got.cpu().numpy(), want)
    pil = video.postprocess_video_webapp(v.cuda())
    assert len(pil) == v.shape[1] and np.array_equal(np.asarray(pil[2]), want[2])
    g = torch.Generator().manual_seed(1)
    big = torch.randn(3, 4, 64, 96, generator=g)
    assert np.array_equal(video.frames_u8(big.cuda(), truncate=True).cpu().numpy(), oframes.frames_u8_webapp(big))
