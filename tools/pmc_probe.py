#!/usr/bin/env python3
"""Launches the two dominant kernels (ffn.0 GEMM at the 5B-c0 shape, self-attention) a few times — the workload for the
rocprofv3 --pmc passes (tools/run_pmc.sh). Inputs are random (not zero) data."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import ops
DEV = "cuda"
L, C, H, FF = 9460, 3072, 24, 14336
bf = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)
a, w, b = bf(L, C), bf(FF, C), torch.randn(FF, device=DEV)
o = torch.empty(L, FF, dtype=torch.bfloat16, device=DEV)
q, k, vt = bf(L, C), bf(L, C), bf(C, (L + 7) // 8 * 8)
oa = torch.empty(L, C, dtype=torch.bfloat16, device=DEV)
xs = torch.randn(L, C, device=DEV); tab = torch.randn(2, 6, C, device=DEV); h = torch.empty(L, C, dtype=torch.bfloat16, device=DEV)
for it in range(4):
    ops.gemm_bf16(a, w, b, o, ops.EPI_BF16_GELU, variant=2)
    ops.gemm_bf16(a, w, b, o, ops.EPI_BF16_GELU, variant=1)
    ops.attn_fwd(q, k, vt, oa, L, L, H)
    ops.adaln_modulate(xs, tab[:, 1], tab[:, 0], 6 * C, None, True, h, 0)
torch.cuda.synchronize()
