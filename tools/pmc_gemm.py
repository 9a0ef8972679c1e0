#!/usr/bin/env python3
"""Workload of tools/run_pmc_gemm.sh: ONE GEMM shape per process (so every rocprofv3 kernel row is one shape), run by the 8-wave 256x256
kernel (variant 2), the one-wave-per-SIMD kernel (variant 3) and the vendor library (torch -> hipBLASLt), a few launches each. Random data.
usage: pmc_gemm.py <M> <N> <K> [epi]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else ops.EPI_BF16
DEV = "cuda"
a = (torch.randn(M, K, device=DEV) * 0.5).to(torch.bfloat16)
w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device=DEV)
bb = b.bfloat16()
o = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
x = torch.randn(M, N, device=DEV) if epi == ops.EPI_RESID else None
for it in range(4):
    for v in (2, 3):
        if epi == ops.EPI_RESID:
            ops.gemm_bf16(a, w, b, x, epi, variant=v)
        else:
            ops.gemm_bf16(a, w, b, o, epi, variant=v)
    if os.environ.get("PMC_NO_LIB") != "1":
        torch.nn.functional.linear(a, w, bb)
torch.cuda.synchronize()
