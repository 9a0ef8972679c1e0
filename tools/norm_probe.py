#!/usr/bin/env python3
"""Timing of the HBM-bound row kernels (adaLN, RMSNorm+RoPE) at the block's shapes, back-to-back launches. A/B of library builds:
YUME_HIP_LIB=yume_amd/lib/exp/libyume_hip_<tag>.so python tools/norm_probe.py (r5 used it for the wave-per-row kernels of commit 307463c:
profiles/r5_norm_wave_ab.json).

Prints one JSON line: per case the average launch time, and the achieved GB/s on the algorithmic bytes."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, reps=200):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def main():
    g = torch.Generator(device=DEV).manual_seed(1)
    out = {"lib": os.environ.get("YUME_HIP_LIB", "product")}
    for name, L, C in (("5b", 9460, 3072), ("14b", 27810, 5120)):
        x = torch.randn((L, C), generator=g, device=DEV)
        tab = torch.randn((2, 6, C), generator=g, device=DEV) * 0.1
        ridx = (torch.arange(L, device=DEV) >= L // 7).to(torch.int32)
        h = torch.empty((L, C), dtype=torch.bfloat16, device=DEV)
        us = timeit(lambda: ops.adaln_modulate(x, tab[:, 1], tab[:, 0], 6 * C, ridx, True, h, 0, 1e-6))
        out[f"adaln_{name}"] = {"us": us, "GBps": 6.0 * L * C / us / 1e3}
        Lp = (L + 63) // 64 * 64
        qk = (torch.randn((Lp, 2 * C), generator=g, device=DEV)).to(torch.bfloat16)
        w = 1 + 0.1 * torch.randn(2 * C, generator=g, device=DEV)
        rope = torch.randn((L, 64, 2), generator=g, device=DEV)
        us = timeit(lambda: ops.rmsnorm_rope(qk[:L], C, 2, w, 1e-6, rope))
        out[f"rmsnorm_rope_qk_{name}"] = {"us": us, "GBps": 8.0 * L * C / us / 1e3}
        us = timeit(lambda: ops.rmsnorm_rope(qk[:L, :C], C, 1, w[:C].contiguous(), 1e-6))
        out[f"rmsnorm_cross_q_{name}"] = {"us": us, "GBps": 4.0 * L * C / us / 1e3}
        del x, qk, h, rope
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
