"""TEST INFRASTRUCTURE — the reference's OWN bf16 deviation at block level, as a committed fixture ("Tolerance guidance").

The real reference block (wan23/modules/model.py:235-316, imported from /root/reference by oracle/ref_import.py) is run on CPU twice on
the same full-width inputs:
  gold : fp32 parameters, no autocast, exact-softmax attention stand-in (the fp32 "gold" every parity test uses);
  bf16 : the same module under torch.autocast("cpu", dtype=torch.bfloat16) with an attention stand-in that follows flash_attention's GPU
         dtype flow (attention.py:56-75,96-130: q, k, v cast to bf16, fp32 softmax statistics, P rounded to bf16 before P·V, bf16 output)
         — i.e. what the reference's own bf16 execution deviates from its fp32 arithmetic.
BASELINE.json configs[0] geometry in the 5B family: one WanAttentionBlock at full width (dim 3072, ffn 14336, 24 heads), L = 2048 tokens
(an [8, 32, 32] latent clip, (1,2,2) patches), 77 text tokens. Inputs and weights come from oracle/fullsize.py::make_block_case (seeded;
the test regenerates them), so the fixture stores only a row sample of both outputs plus the full-tensor statistics:
tests/golden/block_bf16_deviation.pt. The GPU test holds the device block to "within 2x of the reference's own bf16 deviation".

    python oracle/make_golden_bf16dev.py        # build container only (needs /root/reference); ~1 min on 8 cores
"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fullsize, ref_import  # noqa: E402
from yume_amd import synth  # noqa: E402

L, N_TEXT, SEED, N_ROWS = 2048, 77, 41, 96


def flash_dtype_flow_standin(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None, causal=False,
                             window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, version=None):
    """flash_attention()'s signature (attention.py:24-38) and GPU dtype flow: half(x) casts q, k, v to `dtype` (bf16), flash-attn keeps
    scores / statistics in fp32, rounds P to bf16 for the P·V product (fp32 accumulation) and returns bf16 — then `.type(out_dtype)`."""
    assert not causal and dropout_p == 0. and q_lens is None
    out_dtype = q.dtype
    outs = []
    for i in range(q.size(0)):
        lk = int(k_lens[i]) if k_lens is not None else k.size(1)
        qi = q[i].to(dtype).float().transpose(0, 1)
        ki = k[i, :lk].to(dtype).float().transpose(0, 1)
        vi = v[i, :lk].to(dtype).float().transpose(0, 1)
        if q_scale is not None:
            qi = qi * q_scale
        sc = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(q.size(-1))
        s = qi @ ki.transpose(1, 2) * sc
        m = s.amax(dim=-1, keepdim=True)
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)
        o = (p.to(dtype).float() @ vi) / l
        outs.append(o.transpose(0, 1).to(dtype))
    return torch.stack(outs).to(out_dtype)


def reference_block(case):
    mod = ref_import.ref_dit("wan23")
    cfg = case["cfg"]
    blk = mod.WanAttentionBlock(cfg["dim"], cfg["ffn_dim"], cfg["num_heads"], tuple(cfg.get("window_size", (-1, -1))),
                                cfg.get("qk_norm", True), cfg.get("cross_attn_norm", True), cfg.get("eps", 1e-6))
    blk.load_state_dict({k[len("blocks.0."):]: v for k, v in case["sd"].items()}, strict=True)
    return mod, blk.eval().requires_grad_(False)


def run(case, mod, blk, bf16):
    x = case["x"].unsqueeze(0)
    e = case["e6"].unsqueeze(0)
    freqs = case["rope"].unsqueeze(1)                     # flag=True: per-token complex table [L, 1, 64] (rope_apply, model.py:95-104)
    ctx = case["ctx"].unsqueeze(0)
    seq = torch.tensor([case["L"]])
    mod.flash_attention = flash_dtype_flow_standin if bf16 else ref_import.sdpa_standin
    try:
        with torch.no_grad():
            if bf16:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    y = blk(x, e, seq, None, freqs, ctx, None, flag=True)
            else:
                y = blk(x, e, seq, None, freqs, ctx, None, flag=True)
    finally:
        mod.flash_attention = ref_import.sdpa_standin
    return y[0].float()


def stats(got, want, x):
    d = (got.double() - want.double())
    upd = want.double() - x.double()
    return {"rel_l2": (d.norm() / want.double().norm()).item(), "update_rel_l2": (d.norm() / upd.norm()).item(),
            "max_abs": d.abs().max().item(), "ref_rms": want.double().pow(2).mean().sqrt().item()}


def main():
    assert ref_import.available(), "needs the reference tree"
    case = fullsize.make_block_case(synth.CFG_5B, "wan23", L, seed=SEED, n_text=N_TEXT)
    mod, blk = reference_block(case)
    gold = run(case, mod, blk, bf16=False)
    dev = run(case, mod, blk, bf16=True)
    st = stats(dev, gold, case["x"])
    rows = torch.randperm(L, generator=torch.Generator().manual_seed(SEED + 1))[:N_ROWS].sort().values
    fx = dict(L=L, n_text=N_TEXT, seed=SEED, rows=rows, gold_rows=gold[rows].clone(), bf16_rows=dev[rows].clone(),
              x_checksum=float(case["x"].double().sum()), reference_bf16_deviation=st,
              what="reference wan23 WanAttentionBlock on CPU: fp32 gold vs torch.autocast('cpu', bf16) + flash-attn dtype-flow stand-in")
    path = os.path.join(ROOT, "tests", "golden", "block_bf16_deviation.pt")
    torch.save(fx, path)
    print("reference's own bf16 deviation:", st)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
