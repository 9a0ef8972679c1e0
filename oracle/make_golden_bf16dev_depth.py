"""TEST INFRASTRUCTURE — the reference's OWN bf16 deviation at FULL DEPTH, as a committed fixture (SURVEY §8(c):
"errors compound over 30-40 blocks ... full-model tolerances must be measured, not assumed").

oracle/make_golden_bf16dev.py measures ONE real reference block; this script stacks the real reference `WanAttentionBlock`
(wan23/modules/model.py:235-316, imported from /root/reference) 30 times — 30 different full-width 5B blocks (dim 3072, 24 heads,
ffn 14336; hashed synthetic weights, yume_amd.synth.HashedDitStateDict, keys blocks.0 ... blocks.29), L = 2048 tokens, 77 text tokens
— and runs the stack on CPU twice on the same inputs:
  gold : fp32 parameters, no autocast, exact-softmax attention stand-in;
  bf16 : every block under torch.autocast("cpu", dtype=torch.bfloat16) with the flash-attn dtype-flow stand-in (q, k, v and P rounded to
         bf16, fp32 statistics, bf16 output), the residual stream fp32 between blocks — the reference's GPU dtype flow
         (`WanModel.forward` itself cannot run under CPU autocast: its time-embedding assert, model.py:812, sits in an
         `amp.autocast('cuda')` island that is a no-op on CPU; the blocks can).
It stores a row sample of both residual streams after 10, 20 and 30 blocks plus the full-tensor statistics at every depth:
tests/golden/stack_bf16_deviation.pt. The GPU test runs the device blocks on the same inputs and holds the device stack to
"within 2x of the reference's own bf16 deviation" at each stored depth.

    python oracle/make_golden_bf16dev_depth.py        # build container only (needs /root/reference); ~10 min on 8 cores
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_golden_bf16dev as one  # noqa: E402
from oracle import ref_import  # noqa: E402
from yume_amd import synth  # noqa: E402

L, N_TEXT, SEED, N_ROWS, DEPTHS = 2048, 77, 43, 64, (10, 20, 30)


def make_stack_case(n_layers=30):
    """inputs of the stack (seeded CPU generator): residual stream x [L, C], per-token time projection (two distinct rows as on the
    FramePack path), per-token RoPE phases, embedded context. Weights: HashedDitStateDict(seed=SEED), one block at a time."""
    cfg = dict(synth.CFG_5B)
    cfg["num_layers"] = n_layers
    C = cfg["dim"]
    g = torch.Generator().manual_seed(SEED + 1)
    x = torch.randn(L, C, generator=g)
    rows = torch.randn(2, 6, C, generator=g) * 0.1
    n_hist = L // 7
    e6 = torch.cat([rows[0:1].expand(n_hist, 6, C), rows[1:2].expand(L - n_hist, 6, C)]).contiguous()
    ctx = torch.randn(N_TEXT, C, generator=g)
    rope = torch.polar(torch.ones(L, 64, dtype=torch.float64), torch.randn(L, 64, generator=g).double())
    return dict(cfg=cfg, family="wan23", x=x, e6=e6, rope=rope, ctx=ctx, L=L, sd=synth.HashedDitStateDict(cfg, "wan23", SEED))


def block_weights(case, i):
    pre = f"blocks.{i}."
    return {k[len(pre):]: case["sd"][k] for k in case["sd"].keys() if k.startswith(pre)}


def main():
    assert ref_import.available(), "needs the reference tree"
    case = make_stack_case()
    cfg = case["cfg"]
    mod = ref_import.ref_dit("wan23")
    blk = mod.WanAttentionBlock(cfg["dim"], cfg["ffn_dim"], cfg["num_heads"], tuple(cfg.get("window_size", (-1, -1))),
                                cfg.get("qk_norm", True), cfg.get("cross_attn_norm", True), cfg.get("eps", 1e-6)).eval().requires_grad_(False)
    rows = torch.randperm(L, generator=torch.Generator().manual_seed(SEED + 2))[:N_ROWS].sort().values
    xg = xb = case["x"]
    curve, keep = [], {}
    t0 = time.time()
    for i in range(cfg["num_layers"]):
        blk.load_state_dict(block_weights(case, i), strict=True)
        c = dict(case)
        c["x"] = xg
        xg = one.run(c, mod, blk, bf16=False)
        c["x"] = xb
        xb = one.run(c, mod, blk, bf16=True)
        d = (xb.double() - xg.double())
        st = {"depth": i + 1, "rel_l2": (d.norm() / xg.double().norm()).item(),
              "update_rel_l2": (d.norm() / (xg.double() - case["x"].double()).norm()).item(), "max_abs": d.abs().max().item(),
              "ref_rms": xg.double().pow(2).mean().sqrt().item()}
        curve.append(st)
        print(f"[{time.time() - t0:6.0f} s] depth {i + 1}: {st}", flush=True)
        if i + 1 in DEPTHS:
            keep[i + 1] = dict(gold_rows=xg[rows].clone(), bf16_rows=xb[rows].clone())
    fx = dict(L=L, n_text=N_TEXT, seed=SEED, rows=rows, depths=keep, curve=curve, x_checksum=float(case["x"].double().sum()),
              what="30 stacked reference wan23 WanAttentionBlocks on CPU: fp32 gold vs torch.autocast('cpu', bf16) + flash-attn dtype-flow stand-in")
    path = os.path.join(ROOT, "tests", "golden", "stack_bf16_deviation.pt")
    torch.save(fx, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
