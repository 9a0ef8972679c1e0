#!/usr/bin/env python3
"""Per-kernel timing at the real Yume-5B-c0 shapes (L = 9460, C = 3072, 24 heads, ffn 14336) on one MI355X.
Writes gpurun_out/probe.json. Not a test: numbers feed DESIGN.md and kernel tuning."""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import ops  # noqa: E402

DEV = "cuda"
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)


def timeit(fn, warm=2, iters=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def main():
    res = {"device": torch.cuda.get_device_name(0)}
    L, C, H, FF = 9460, 3072, 24, 14336
    bf = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)
    variants = [int(v) for v in os.environ.get("YUME_GEMM_VARIANTS", "1,2,0").split(",")]
    # ---- GEMMs
    shapes = {"qkv": (L, 3 * C, C), "o": (L, C, C), "ffn1": (L, FF, C), "ffn2": (L, C, FF), "sq4096": (4096, 4096, 4096),
              "sq8192": (8192, 8192, 8192)}
    for name, (M, N, K) in shapes.items():
        a, w, b = bf(M, K), bf(N, K), torch.randn(N, device=DEV)
        o = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        for v in variants:
            ms = timeit(lambda: ops.gemm_bf16(a, w, b, o, ops.EPI_BF16, variant=v))
            res[f"gemm_{name}_v{v}"] = {"ms": ms, "tflops": 2 * M * N * K / ms / 1e9}
            print(f"gemm {name} v{v} {M}x{N}x{K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TF", flush=True)
        del a, w, o
    # epilogue variants at the o-proj shape
    M, N, K = L, C, C
    a, w, b = bf(M, K), bf(N, K), torch.randn(N, device=DEV)
    x = torch.randn(M, N, device=DEV)
    tab = torch.randn(2, 6, N, device=DEV)
    idx = (torch.arange(M, device=DEV) >= 2420).to(torch.int32)
    ms = timeit(lambda: ops.gemm_bf16(a, w, b, x, ops.EPI_RESID, gate=tab[:, 2], gate_stride=6 * N, row_idx=idx))
    res["gemm_o_resid"] = {"ms": ms, "tflops": 2 * M * N * K / ms / 1e9}
    print(f"gemm o resid: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF", flush=True)
    qk = torch.empty(M, 2 * C, dtype=torch.bfloat16, device=DEV)
    vt = torch.empty(C, (M + 7) // 8 * 8, dtype=torch.bfloat16, device=DEV)
    w3 = bf(3 * C, C)
    b3 = torch.randn(3 * C, device=DEV)
    ms = timeit(lambda: ops.gemm_bf16(a, w3, b3, qk, ops.EPI_BF16_SPLITT, out_t=vt, n_split=2 * C))
    res["gemm_qkv_splitt"] = {"ms": ms, "tflops": 2 * M * 3 * C * K / ms / 1e9}
    print(f"gemm qkv splitT: {ms:.3f} ms {2*M*3*C*K/ms/1e9:.0f} TF", flush=True)
    # ---- attention
    for (Lq, Lk, tag) in ((L, L, "self"), (L, 512, "cross"), (23460, 23460, "self14b")):
        Hh = 40 if tag == "self14b" else H
        Cc = Hh * 128
        q, k = bf(Lq, Cc), bf(Lk, Cc)
        vtt = bf(Cc, (Lk + 7) // 8 * 8)
        o = torch.empty(Lq, Cc, dtype=torch.bfloat16, device=DEV)
        fl = 4 * Lq * Lk * Cc
        for av in [int(v) for v in os.environ.get("YUME_ATTN_VARIANTS", "1,2,4,0").split(",")]:
            ms = timeit(lambda: ops.attn_fwd(q, k, vtt, o, Lq, Lk, Hh, variant=av), warm=1, iters=3)
            res[f"attn_{tag}_v{av}"] = {"ms": ms, "tflops": fl / ms / 1e9}
            print(f"attn {tag} v{av} Lq={Lq} Lk={Lk} H={Hh}: {ms:.3f} ms {fl/ms/1e9:.0f} TF", flush=True)
        del q, k, vtt, o
    # ---- HBM-bound kernels
    xs = torch.randn(L, C, device=DEV)
    h = torch.empty(L, C, dtype=torch.bfloat16, device=DEV)
    ms = timeit(lambda: ops.adaln_modulate(xs, tab[:, 1], tab[:, 0], 6 * C, idx, True, h, 0))
    res["adaln"] = {"ms": ms, "gbps": (4 * L * C + 2 * L * C) / ms / 1e6}
    print(f"adaln: {ms:.4f} ms {(6*L*C)/ms/1e6:.0f} GB/s", flush=True)
    rope = torch.randn(L, 64, 2, device=DEV)
    nw = torch.ones(2 * C, device=DEV)
    ms = timeit(lambda: ops.rmsnorm_rope(qk, C, 2, nw, 1e-6, rope))
    res["rmsnorm_rope"] = {"ms": ms, "gbps": (2 * 2 * 2 * L * C) / ms / 1e6}
    print(f"rmsnorm_rope: {ms:.4f} ms {(8*L*C)/ms/1e6:.0f} GB/s", flush=True)
    with open(os.path.join(OUT, "probe.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
