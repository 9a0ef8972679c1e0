#!/usr/bin/env python3
"""Clock / power while one kernel runs back to back for a few seconds (rocm-smi sampled from a side thread). Answers whether a
kernel's ceiling is the chip's power / clock management rather than its own schedule. Writes gpurun_out/power_probe.json."""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import ops  # noqa: E402

DEV = "cuda"


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showuse", "--json"], capture_output=True, text=True, timeout=5).stdout
            out.append((time.time(), json.loads(txt)))
        except Exception as ex:  # noqa: BLE001
            out.append((time.time(), {"error": str(ex)[:200]}))
        time.sleep(0.05)


def run(name, fn, flops, seconds=3.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples))
    th.start()
    t0 = time.time()
    n = 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    stop.set()
    th.join()
    vals = {}
    for _, d in samples:
        card = d.get("card0", {})
        for k, v in card.items():
            m = re.search(r"[-+]?\d+(\.\d+)?", str(v))
            if m:
                vals.setdefault(k, []).append(float(m.group(0)))
    summ = {k: {"mean": sum(v) / len(v), "max": max(v), "min": min(v)} for k, v in vals.items()}
    print(f"{name}: {ms:.4f} ms/launch, {flops / ms / 1e9:.0f} TF, {len(samples)} samples", flush=True)
    for k, v in summ.items():
        print(f"    {k}: mean {v['mean']:.1f} min {v['min']:.1f} max {v['max']:.1f}", flush=True)
    return {"ms": ms, "tflops": flops / ms / 1e9, "smi": summ}


def main():
    res = {}
    bf = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)
    idle = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
    print(idle, flush=True)
    M = N = K = 8192
    a, w, b = bf(M, K), bf(N, K) * K ** -0.5, torch.randn(N, device=DEV)
    o = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    res["gemm_8192"] = run("gemm 8192^3 (own 256x256 kernel)", lambda: ops.gemm_bf16(a, w, b, o, ops.EPI_BF16), 2 * M * N * K)
    bb = b.bfloat16()
    res["gemm_8192_lib"] = run("gemm 8192^3 (library)", lambda: torch.nn.functional.linear(a, w, bb), 2 * M * N * K)
    # same kernel on a quarter of the chip's worth of tiles: 256 tiles = one round
    a2, w2 = bf(4096, K), bf(4096, K) * K ** -0.5
    o2 = torch.empty(4096, 4096, dtype=torch.bfloat16, device=DEV)
    res["gemm_4096x4096x8192"] = run("gemm 4096x4096x8192 (one round of 256 tiles)", lambda: ops.gemm_bf16(a2, w2, None, o2, ops.EPI_BF16, variant=2), 2 * 4096 * 4096 * K)
    a3, w3 = bf(4096, K), bf(2048, K) * K ** -0.5
    o3 = torch.empty(4096, 2048, dtype=torch.bfloat16, device=DEV)
    res["gemm_4096x2048x8192"] = run("gemm 4096x2048x8192 (128 tiles: half the CUs)", lambda: ops.gemm_bf16(a3, w3, None, o3, ops.EPI_BF16, variant=2), 2 * 4096 * 2048 * K)
    L, H = 9460, 24
    q, k = bf(L, H * 128), bf(L, H * 128)
    vt = bf(H * 128, (L + 7) // 8 * 8)
    oo = torch.empty(L, H * 128, dtype=torch.bfloat16, device=DEV)
    res["attn_5b"] = run("self-attention 9460 x 9460 x 24", lambda: ops.attn_fwd(q, k, vt, oo, L, L, H), 4 * L * L * 128 * H)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "power_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
