#!/usr/bin/env python3
"""Workload of a rocprofv3 --pmc pass: the dominant Wan2.2-decoder convolution (3x3x3 256->256 @4x352x640) a few times. Random data."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import vae_ops as V
DEV = "cuda"
ci, co, T, H, W, k = 256, 256, 4, 352, 640, (3, 3, 3)
x = (torch.randn(T, H, W, ci, device=DEV) * 0.5).to(torch.bfloat16)
cache = (torch.randn(2, H, W, ci, device=DEV) * 0.5).to(torch.bfloat16)
K = 27 * ci
w = (torch.randn(co, K, device=DEV) * K ** -0.5).to(torch.bfloat16)
b = torch.zeros(co, device=DEV)
zero = torch.zeros(64, dtype=torch.bfloat16, device=DEV)
out = torch.empty(T, H, W, co, dtype=torch.bfloat16, device=DEV)
for _ in range(4):
    V.conv3d_cl(x, cache, w, b, co, k, (1, 1, 1), (2, 1, 1), False, out, V.EPI_BF16, zero_page=zero)
torch.cuda.synchronize()
