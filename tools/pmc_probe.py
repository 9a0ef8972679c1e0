#!/usr/bin/env python3
"""Launches the kernels that own the 5B denoise step and the VAE decode a few times — the workload of the rocprofv3 --pmc passes
(tools/run_pmc.sh): self-attention (automatic selection), the QKV / o-proj / ffn.0 / ffn.2 GEMMs at the 5B-c0 shapes with their
epilogues, adaLN, and the dominant Wan2.2-decoder convolution. Inputs are random (not zero) data."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import ops, vae_ops as V
DEV = "cuda"
L, C, H, FF = 9460, 3072, 24, 14336
bf = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)
a, w1, b1 = bf(L, C), bf(FF, C) * 0.05, torch.randn(FF, device=DEV)
ff = torch.empty(L, FF, dtype=torch.bfloat16, device=DEV)
w2, b2 = bf(C, FF) * 0.02, torch.randn(C, device=DEV)
wo = bf(C, C) * 0.05
wqkv, bqkv = bf(3 * C, C) * 0.05, torch.randn(3 * C, device=DEV)
qk = torch.empty(L, 2 * C, dtype=torch.bfloat16, device=DEV)
vt = torch.empty(C, (L + 7) // 8 * 8, dtype=torch.bfloat16, device=DEV)
q, k = (bf(L, C).float() * (2.0 * 0.12753)).to(torch.bfloat16), bf(L, C) * 2.0     # q as the engine hands it over: times scale * log2(e)
oa = torch.empty(L, C, dtype=torch.bfloat16, device=DEV)
xs = torch.randn(L, C, device=DEV)
tab = torch.randn(2, 6, C, device=DEV)
ridx = (torch.arange(L, device=DEV) % 2).to(torch.int32)
h = torch.empty(L, C, dtype=torch.bfloat16, device=DEV)
# Wan2.2 decoder: 3x3x3 256->256 at 4x352x640 (23 % of the decode FLOPs)
cx = bf(4, 352, 640, 256); cc = bf(2, 352, 640, 256)
cw = (torch.randn(256, 27 * 256, device=DEV) * (27 * 256) ** -0.5).to(torch.bfloat16)
cb = torch.zeros(256, device=DEV); co = torch.empty(4, 352, 640, 256, dtype=torch.bfloat16, device=DEV)
zero = torch.zeros(64, dtype=torch.bfloat16, device=DEV)
for it in range(3):
    ops.gemm_bf16(a, wqkv, bqkv, qk, ops.EPI_BF16_SPLITT, out_t=vt, n_split=2 * C)
    ops.attn_fwd(q, k, vt, oa, L, L, H, q_prescaled=True)
    ops.gemm_bf16(oa, wo, b2, xs, ops.EPI_RESID, gate=tab[:, 2], gate_stride=6 * C, row_idx=ridx)
    ops.adaln_modulate(xs, tab[:, 1], tab[:, 0], 6 * C, ridx, True, h, 0)
    ops.gemm_bf16(h, w1, b1, ff, ops.EPI_BF16_GELU)
    ops.gemm_bf16(ff, w2, b2, xs, ops.EPI_RESID, gate=tab[:, 5], gate_stride=6 * C, row_idx=ridx)
    V.conv3d_cl(cx, cc, cw, cb, 256, (3, 3, 3), (1, 1, 1), (2, 1, 1), False, co, V.EPI_BF16, zero_page=zero)
torch.cuda.synchronize()
