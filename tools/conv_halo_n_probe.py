#!/usr/bin/env python3
"""Timing of the 96 / 160-channel 3x3x3 convolutions (conv_halo_n.hpp) at production frame sizes on one MI355X.
   YUME_CONV_HALO_N=0 -> the r5 path (generic-loader 128x128 kernel); YUME_CONV_HALO_NW=4|8 -> one / two waves per SIMD."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import vae_ops as V
DEV = "cuda"
def timeit(fn, warm=2, iters=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
zero = torch.zeros(64, dtype=torch.bfloat16, device=DEV)
shapes = [("3x3x3 96->96 @16x544x960 (Wan2.1 dec/enc level 0)", 96, 16, 544, 960),
          ("3x3x3 96->96 @4x544x960", 96, 4, 544, 960),
          ("3x3x3 160->160 @8x352x640 (Wan2.2 enc level 0)", 160, 8, 352, 640)]
for name, c, T, H, W in shapes:
    x = (torch.randn(T, H, W, c, device=DEV) * 0.5).to(torch.bfloat16)
    cache = (torch.randn(2, H, W, c, device=DEV) * 0.5).to(torch.bfloat16)
    K = 27 * c
    Kp = (K + 63) // 64 * 64
    w = (torch.randn(c, Kp, device=DEV) * K ** -0.5).to(torch.bfloat16)
    b = torch.zeros(c, device=DEV)
    out = torch.empty(T, H, W, c, dtype=torch.bfloat16, device=DEV)
    ms = timeit(lambda: V.conv3d_cl(x, cache, w, b, c, (3, 3, 3), (1, 1, 1), (2, 1, 1), False, out, V.EPI_BF16, zero_page=zero))
    fl = 2.0 * T * H * W * c * K
    print(f"{name:52s} {ms:8.3f} ms  {fl/ms/1e9:7.0f} TF", flush=True)
