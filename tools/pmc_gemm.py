#!/usr/bin/env python3
"""Workload of tools/run_pmc_gemm.sh: ONE GEMM of the 5B block per process (so every rocprofv3 kernel row is one shape AND one epilogue),
launched as the engine launches it (yume_amd/dit.py), a few times. Random data.
usage: pmc_gemm.py <qkv|o|cross_q|cross_o|ffn0|ffn2|M N K> [variant]      (variant 0 = automatic, 2 = 8-wave kernel, 3 = one wave per SIMD)
With PMC_WITH_LIB=1 the vendor library (torch -> hipBLASLt) runs the same product with a plain bf16 epilogue as a yardstick."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import ops
DEV = "cuda"
L, C, FF = 9460, 3072, 14336
name = sys.argv[1]
shapes = {"qkv": (L, 3 * C, C), "o": (L, C, C), "cross_q": (L, C, C), "cross_o": (L, C, C), "ffn0": (L, FF, C), "ffn2": (L, C, FF)}
if name in shapes:
    M, N, K = shapes[name]
    variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
else:
    M, N, K = (int(v) for v in sys.argv[1:4])
    variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    name = "plain"
a = (torch.randn(M, K, device=DEV) * 0.5).to(torch.bfloat16)
w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device=DEV)
tab = torch.randn(2, 6, C, device=DEV)
ridx = (torch.arange(M, device=DEV) % 2).to(torch.int32)
if name == "qkv":
    out = torch.empty(M, 2 * C, dtype=torch.bfloat16, device=DEV)
    vt = torch.empty(C, (M + 7) // 8 * 8, dtype=torch.bfloat16, device=DEV)
    call = lambda: ops.gemm_bf16(a, w, b, out, ops.EPI_BF16_SPLITT, out_t=vt, n_split=2 * C, variant=variant)
elif name in ("o", "ffn2"):
    x = torch.randn(M, N, device=DEV)
    call = lambda: ops.gemm_bf16(a, w, b, x, ops.EPI_RESID, gate=tab[:, 2], gate_stride=6 * C, row_idx=ridx, variant=variant)
elif name == "cross_o":
    x = torch.randn(M, N, device=DEV)
    call = lambda: ops.gemm_bf16(a, w, b, x, ops.EPI_RESID, variant=variant)
elif name == "ffn0":
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    call = lambda: ops.gemm_bf16(a, w, b, out, ops.EPI_BF16_GELU, variant=variant)
else:
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    call = lambda: ops.gemm_bf16(a, w, b, out, ops.EPI_BF16, variant=variant)
bb = b.bfloat16()
for it in range(4):
    call()
    if os.environ.get("PMC_WITH_LIB") == "1":
        torch.nn.functional.linear(a, w, bb)
torch.cuda.synchronize()
