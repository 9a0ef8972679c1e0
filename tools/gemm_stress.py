#!/usr/bin/env python3
"""Randomised cross-check of yume_gemm_bf16's automatic kernel selection (256x256 / 128x128 / row split) against the explicit
kernels for every epilogue. Not a test file: prints mismatches."""
import random
import sys

import torch

sys.path.insert(0, ".")
from yume_amd import ops  # noqa: E402

DEV = "cuda"


def main():
    random.seed(2)
    bad = n = 0
    for _ in range(40):
        M = random.choice([1, 77, 255, 257, 1000, 4100, 9460, 9461, 8192 + 300, 12000])
        N = random.choice([128, 192, 3072, 4096, 7168, 9216])
        K = random.choice([64, 128, 512, 1024])
        a = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
        w = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
        bias = torch.randn(N, device=DEV)
        x = torch.randn(M, N, device=DEV)
        tab = torch.randn(2, 6, N, device=DEV)
        idx = (torch.arange(M, device=DEV) % 2).to(torch.int32)
        Mp = (M + 7) // 8 * 8
        ns = (N // 2) // 256 * 256
        res = {}
        for var in (1, 0):
            o32 = ops.gemm_bf16(a, w, bias, torch.empty(M, N, device=DEV), ops.EPI_F32, variant=var)
            og = ops.gemm_bf16(a, w, bias, torch.empty(M, N, dtype=torch.bfloat16, device=DEV), ops.EPI_BF16_GELU, variant=var)
            xr = ops.gemm_bf16(a, w, bias, x.clone(), ops.EPI_RESID, gate=tab[:, 2], gate_stride=6 * N, row_idx=idx, variant=var)
            qk = torch.empty(M, max(ns, 4), dtype=torch.bfloat16, device=DEV)
            vt = torch.zeros(N - ns, Mp, dtype=torch.bfloat16, device=DEV)
            if ns:
                ops.gemm_bf16(a, w, bias, qk, ops.EPI_BF16_SPLITT, out_t=vt, n_split=ns, variant=var)
            res[var] = (o32, og.float(), xr, qk.float() if ns else o32, vt.float())
        for i, (p, q) in enumerate(zip(res[1], res[0])):
            n += 1
            tol = 3e-5 if i in (0, 2) else 2.0 ** -7
            d = (p - q).abs().max().item()
            if not torch.isfinite(q).all() or d > tol * max(1.0, p.abs().max().item()):
                bad += 1
                print("MISMATCH", (M, N, K), "output", i, d)
    print("checks", n, "bad", bad)


if __name__ == "__main__":
    main()
