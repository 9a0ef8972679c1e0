#!/usr/bin/env python3
"""Multi-step drift of the full-depth 5B denoise chain, device vs CPU oracle ("a 4-step 5B chain drift", the shipped 5B default is
4 steps: scripts/inference/sample_5b.sh:18). Not a test — a record for profiles/.

Case oracle/step_job.py::5b_chain: the full 30-block Yume-5B (hashed synthetic weights), latent [48,13,22,40] (a quarter of the 704x1280
clip's area: L = 2380, so that four sequential fp32 CPU forwards cost ~3 min instead of 14), 4 Euler steps of the shift-7 schedule with
clean history (sample_5b.py:960-1034). Both chains start from the same latent; each runs on its OWN previous result, so the figure is the
accumulated difference after k steps, not a per-step error.

    python tools/chain_drift.py [--threads 32]  ->  gpurun_out/chain_drift.json

Round 5: `--case 5b --gold cuda` runs the FULL-AREA case (latent [48,13,44,80], L = 9460) over the whole 50-step shift-7
schedule of BASELINE configs[1] with the gold chain on the device gold (oracle/devgold.py: oracle/dit.py on the GPU in fp32, ~2 s per
forward instead of 4 minutes) -> gpurun_out/chain_drift_5b_50steps.json. The device gold is proven against the CPU oracle at this size in
tests/test_zz_full_step_gpu.py."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import step_job  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--case", default="5b_chain", choices=["5b_chain", "5b"])
    ap.add_argument("--steps", type=int, default=0, help="chain length (default: the case's schedule length)")
    ap.add_argument("--gold", default="cpu", choices=["cpu", "cuda"], help="where the gold chain's oracle forwards run")
    a = ap.parse_args()
    name = a.case
    c = step_job.CASES[name]
    lfz, S = c["lfz"], (a.steps or c["steps"])
    model = step_job.build_device_model(name, "cuda")
    assert step_job.weights_agree(name, model)
    lat0 = step_job.make_inputs(name)["latent"]
    hist = lat0[:, :-lfz]
    ld, lo = lat0.clone(), lat0.clone()
    rec = []
    for i in range(S):
        pd = step_job.device_forward(name, model, "cond", latent=ld, i=i).cpu()
        po, secs, _ = step_job.oracle_forward(name, "cond", latent=lo, i=i, threads=a.threads, device=None if a.gold == "cpu" else a.gold)
        xd, xo = step_job.euler(name, ld, pd, i), step_job.euler(name, lo, po, i)
        ld, lo = torch.cat([hist, xd], dim=1), torch.cat([hist, xo], dim=1)
        r = {"step": i + 1, "pred": step_job.stats(pd, po), "latent": step_job.stats(xd, xo), "oracle_seconds": secs}
        rec.append(r)
        print(json.dumps(r), flush=True)
    out = {"case": name, "L": step_job.seq_len(name).seq_len, "steps": S, "gold": "oracle/dit.py fp32 on " + a.gold,
           "what": __doc__.split("\n\n")[1], "chain": rec}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    fn = "chain_drift.json" if (name == "5b_chain" and a.gold == "cpu") else f"chain_drift_{name}_{S}steps.json"
    with open(os.path.join(ROOT, "gpurun_out", fn), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
