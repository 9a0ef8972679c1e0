#!/usr/bin/env python3
"""Implicit-GEMM conv timing at the dominant Wan2.2-decoder shapes on one MI355X."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import vae_ops as V
DEV = "cuda"
def timeit(fn, warm=1, iters=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
zero = torch.zeros(64, dtype=torch.bfloat16, device=DEV)
shapes = [("3x3x3 256->256 @4x352x640", 256, 256, 4, 352, 640, (3, 3, 3), False),
          ("3x3x3 512->512 @4x176x320", 512, 512, 4, 176, 320, (3, 3, 3), False),
          ("3x3x3 1024->1024 @2x88x160", 1024, 1024, 2, 88, 160, (3, 3, 3), False),
          ("3x3x3 1024->512 @4x176x320", 1024, 512, 4, 176, 320, (3, 3, 3), False),
          ("1x3x3 up2x 1024->1024 @2x88x160->176x320", 1024, 1024, 2, 88, 160, (1, 3, 3), True),
          ("3x3x3 160->160 @4x352x640 (enc)", 160, 160, 4, 352, 640, (3, 3, 3), False),
          ("3x3x3 256->12 @4x352x640 (head)", 256, 12, 4, 352, 640, (3, 3, 3), False)]
for name, ci, co, T, H, W, k, ups in shapes:
    x = (torch.randn(T, H, W, ci, device=DEV) * 0.5).to(torch.bfloat16)
    cache = (torch.randn(2, H, W, ci, device=DEV) * 0.5).to(torch.bfloat16) if k[0] == 3 else None
    K = k[0] * k[1] * k[2] * ci
    Kp = (K + 63) // 64 * 64
    w = (torch.randn(co, Kp, device=DEV) * K ** -0.5).to(torch.bfloat16)
    b = torch.zeros(co, device=DEV)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    out = torch.empty(T, Ho, Wo, (co + 7) // 8 * 8, dtype=torch.bfloat16, device=DEV)
    ms = timeit(lambda: V.conv3d_cl(x, cache, w, b, co, k, (1, 1, 1), (k[0] - 1, 1, 1), ups, out, V.EPI_BF16, zero_page=zero))
    fl = 2.0 * T * Ho * Wo * co * K
    print(f"{name:48s} {ms:8.3f} ms  {fl/ms/1e9:7.0f} TF", flush=True)
