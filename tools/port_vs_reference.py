#!/usr/bin/env python3
"""How bench.py's `cpu_baseline` (kind "port": oracle/dit.py) relates to the REFERENCE's own modules on the same host.

The GPU box has no reference tree, so the timed CPU leg there is the pinned restatement. Here, in the build container where /root/reference
exists, the real `WanAttentionBlock` (wan23/modules/model.py:235-316, imported by oracle/ref_import.py) and oracle.dit.block_forward run the
SAME full-width block (dim 3072, ffn 14336, 24 heads; L = 2048 tokens + 77 text tokens = BASELINE configs[0]'s geometry in the 5B family;
fp32, no autocast) on identical inputs, alternating, with the same fp32 SDPA attention bound into both (the reference's flash-attn call cannot
run on a CPU; oracle/fullsize.py::attention_fp32 is what the bench's CPU leg uses). The ratio reference / port of the median times goes into
profiles/r5_cpu_port_vs_reference.json; bench.py attaches it to `cpu_baseline` as `reference_over_port`.

    python tools/port_vs_reference.py        # build container only; ~2 min on 8 cores
"""
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fullsize, ref_import  # noqa: E402
from oracle import make_golden_bf16dev as mk  # noqa: E402
from yume_amd import synth  # noqa: E402

L, N_TEXT, REPS = 2048, 77, 5


def sdpa_fp32(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None, causal=False, window_size=(-1, -1),
              deterministic=False, dtype=torch.bfloat16, version=None):
    """flash_attention()'s signature (attention.py:24-38) on oracle/fullsize.py::attention_fp32 — the attention of the bench's CPU leg."""
    outs = []
    for i in range(q.size(0)):
        lk = int(k_lens[i]) if k_lens is not None else k.size(1)
        outs.append(fullsize.attention_fp32(q[i].float(), k[i, :lk].float(), v[i, :lk].float()))
    return torch.stack(outs).to(q.dtype)


def main():
    assert ref_import.available(), "needs the reference tree (/root/reference)"
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    case = fullsize.make_block_case(synth.CFG_5B, "wan23", L, seed=3, n_text=N_TEXT)
    mod, blk = mk.reference_block(case)
    x, e, freqs, ctx, seq = case["x"].unsqueeze(0), case["e6"].unsqueeze(0), case["rope"].unsqueeze(1), case["ctx"].unsqueeze(0), torch.tensor([L])
    mod.flash_attention = sdpa_fp32

    def run_ref():
        with torch.no_grad():
            return blk(x, e, seq, None, freqs, ctx, None, flag=True)[0]

    try:
        y_ref = run_ref()
        y_port, _ = fullsize.run_block_oracle(case)
        diff = ((y_ref.double() - y_port.double()).norm() / y_ref.double().norm()).item()
        t_ref, t_port = [], []
        for _ in range(REPS):
            t0 = time.perf_counter()
            run_ref()
            t_ref.append(time.perf_counter() - t0)
            _, dt = fullsize.run_block_oracle(case)
            t_port.append(dt)
    finally:
        mod.flash_attention = ref_import.sdpa_standin
    mr, mp = statistics.median(t_ref), statistics.median(t_port)
    out = {"reference_over_port": mr / mp, "reference_s": mr, "port_s": mp, "reference_runs_s": t_ref, "port_runs_s": t_port, "threads": threads,
           "host_threads": os.cpu_count(), "outputs_rel_l2": diff,
           "what": f"one full-width 5B WanAttentionBlock, L={L} + {N_TEXT} text tokens, fp32, the same fp32 SDPA attention in both: the REAL reference module "
                   "(wan23/modules/model.py:235-316) vs oracle.dit.block_forward, alternating, median of " + str(REPS)}
    path = os.path.join(ROOT, "profiles", "r5_cpu_port_vs_reference.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
