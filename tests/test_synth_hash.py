"""yume_amd.synth.hashed_uniform: the device-independent weight rule of the full-depth parity cases (oracle/step_job.py on the host,
fill_module_hashed_ on the GPU) against a numpy uint64 evaluation of the same hash, chunk-size independence and the lazy mapping."""
import zlib

import numpy as np
import torch

from yume_amd import synth


def _ref(key, n, seed, a):
    M = np.uint64
    base = M((((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF) * 0xD1B54A32D192ED03) & ((1 << 64) - 1))
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = i * M(0x9E3779B97F4A7C15) + base
        x = (x ^ (x >> M(30))) * M(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> M(27))) * M(0x94D049BB133111EB)
        x = x ^ (x >> M(31))
    u = ((x >> M(41)) & M((1 << 23) - 1)).astype(np.float32)
    u = u * np.float32(2.0 ** -22)
    u = u + np.float32(2.0 ** -23 - 1.0)
    return u * np.float32(a)


def test_hashed_uniform_matches_numpy_uint64_and_is_chunk_independent():
    key, seed, a = "blocks.3.ffn.0.weight", 11, 0.0173
    want = _ref(key, 700_001, seed, a)
    for chunk in (None, 1 << 10, 700_001, 1 << 22):
        got = synth.hashed_uniform(key, (700_001,), seed, a, chunk=chunk)
        assert np.array_equal(got.numpy(), want), chunk
    assert abs(float(want.mean())) < 1e-4 and abs(float(want.std()) / (a / 3 ** 0.5) - 1) < 5e-3
    assert float(np.abs(want).max()) < a
    assert not np.array_equal(_ref(key, 1000, seed + 1, a), want[:1000])


def test_hashed_state_dict_is_lazy_and_keyed():
    cfg = synth.tiny_cfg("wan23", layers=2)
    sd = synth.HashedDitStateDict(cfg, "wan23", seed=4)
    assert "blocks.1.ffn.0.weight" in sd and "blocks.2.ffn.0.weight" not in sd
    w = sd["blocks.1.ffn.0.weight"]
    assert w.shape == (cfg["ffn_dim"], cfg["dim"]) and torch.equal(w, sd["blocks.1.ffn.0.weight"])
    assert not torch.equal(w, sd["blocks.0.ffn.0.weight"])
    assert sd.get("blocks.0.norm3.weight") is not None and sd.get("nope") is None
    # small tensors follow make_tensor (sequential generator); the big matrices follow the hash
    assert torch.equal(sd["blocks.0.modulation"], synth.make_tensor("blocks.0.modulation", (1, 6, cfg["dim"]), 4, cfg["dim"]))


@torch.no_grad()
def test_gpu_values_equal_host_values():
    if not torch.cuda.is_available():
        import pytest
        pytest.skip("no GPU here; the full-depth GPU tests assert this on the box")
    a = synth.hashed_uniform("k", (3, 100_003), 5, 0.5)
    b = synth.hashed_uniform("k", (3, 100_003), 5, 0.5, device="cuda")
    assert torch.equal(a, b.cpu())
