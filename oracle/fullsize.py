"""Full-size live-block cases for the parity tests and bench.py's cpu_baseline leg (TEST INFRASTRUCTURE, like the rest of oracle/).

One real (non-identity) full-width DiT block at the sequence length of BASELINE.json configs[1] (5B, L = 9460) or
configs[2] (14B, L = 27810) is run through oracle/dit.py::block_forward — the restatement pinned to the reference in
tests/test_oracle_dit.py. The only change at this size is the attention inner product: the pinned oracle forms the full
fp64 score matrix of all heads at once (17 GB at L = 9460, 250 GB at 27810); here it is evaluated head by head in fp32
(torch SDPA, exact softmax, no mask). That changes the CPU result by ~1e-6 relative, three orders below the stated 1e-2
parity tolerance of the bf16 device path.
"""
import time

import torch
import torch.nn.functional as F

from . import dit as odit


def attention_fp32(q, k, v, heads_per_chunk=4):
    """attention.py:56-130 semantics (softmax(q k^T / sqrt(D)) v, all keys), q/k/v [L, N, D] fp32, a few heads at a time."""
    outs = []
    for h0 in range(0, q.shape[1], heads_per_chunk):
        sl = slice(h0, h0 + heads_per_chunk)
        outs.append(F.scaled_dot_product_attention(q[:, sl].transpose(0, 1), k[:, sl].transpose(0, 1), v[:, sl].transpose(0, 1)).transpose(0, 1))
    return torch.cat(outs, dim=1)


def make_block_case(cfg, family, L, seed=0, n_text=512):
    """weights of block 0 of a num_layers=1 model (yume_amd.synth generator = the one the device model is filled from) and the
    block's inputs: residual stream x [L, C], time projection e6 ([L, 6, C] per token for the 5B family, two distinct rows
    as on the FramePack path; [6, C] for 14B), per-token RoPE phases [L, 64] complex128, embedded context [(257 +) n_text, C]."""
    from yume_amd import synth
    c1 = dict(cfg)
    c1["num_layers"] = 1
    sd = {k: v for k, v in synth.make_dit_state_dict(c1, family, seed=seed, pyramid=()).items() if k.startswith("blocks.0.")}
    C = cfg["dim"]
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(L, C, generator=g)
    if family == "wan23":
        rows = torch.randn(2, 6, C, generator=g) * 0.1
        n_hist = L // 7
        e6 = torch.cat([rows[0:1].expand(n_hist, 6, C), rows[1:2].expand(L - n_hist, 6, C)]).contiguous()
    else:
        e6 = torch.randn(6, C, generator=g) * 0.1
    ctx = torch.randn((257 if family == "wan" else 0) + n_text, C, generator=g)
    rope = torch.polar(torch.ones(L, 64, dtype=torch.float64), torch.randn(L, 64, generator=g).double())
    return dict(cfg=c1, family=family, sd=sd, x=x, e6=e6, rope=rope, ctx=ctx, L=L)


def run_block_oracle(case, device=None):
    """-> (x_out fp32 [L, C] on the host, seconds). device None: on the host cores; "cuda": the same oracle.dit.block_forward on the GPU in
    fp32 (oracle/devgold.py — for the 14B block at L = 27810, 90 s of host time otherwise)."""
    if device is not None:
        from . import devgold
        dev = torch.device(device)
        sd = {k: v.to(dev) for k, v in case["sd"].items()}
        with torch.no_grad(), devgold.on_device(dev):
            t0 = time.time()
            y = odit.block_forward(sd, "blocks.0.", case["x"].to(dev), case["e6"].to(dev), case["rope"].to(dev), case["ctx"].to(dev),
                                   case["cfg"], case["family"]).cpu()
        return y, time.time() - t0
    orig = odit.attention
    odit.attention = attention_fp32
    try:
        t0 = time.time()
        with torch.no_grad():
            y = odit.block_forward(case["sd"], "blocks.0.", case["x"], case["e6"], case["rope"], case["ctx"], case["cfg"], case["family"])
        return y, time.time() - t0
    finally:
        odit.attention = orig


def run_block_device(case, model, dev="cuda"):
    """the same block through the drop-in's engine (model: yume_amd WanModel with num_layers=1 holding case['sd'])."""
    rope_cs = torch.stack([case["rope"].real, case["rope"].imag], dim=-1).to(torch.float32)
    return model.engine.block_forward(0, case["x"].to(dev), case["e6"].to(dev), rope_cs.to(dev), case["ctx"].to(dev),
                                      n_img=257 if case["family"] == "wan" else 0).cpu()


def build_block_model(case, dev="cuda"):
    import torch.nn as nn  # noqa: F401
    if case["family"] == "wan23":
        from yume_amd.wan23.modules.model import WanModel
    else:
        from yume_amd.wan.modules.model import WanModel
    with torch.device(dev):
        m = WanModel(**case["cfg"])
    missing, unexpected = m.load_state_dict(case["sd"], strict=False)
    assert not unexpected and not [k for k in missing if k.startswith("blocks.")]
    return m.eval().requires_grad_(False)


def parity(got, want):
    d = (got.double() - want.double())
    return {"rel_l2": (d.norm() / want.double().norm()).item(), "max_abs": d.abs().max().item(),
            "ref_rms": want.double().pow(2).mean().sqrt().item()}
