#!/usr/bin/env python3
"""conv_trace.py <libyume_hip_trace_conv.so> <outdir>: runs the dominant Wan2.2-decoder convolutions (tools/conv_probe.py's shapes) on a
-DYUME_TRACE experiment build of conv3d.hip (tools/build_variant.sh trace_conv conv3d.hip -DYUME_TRACE) and dumps the per-workgroup stamps
of the last launch of each (csrc/trace.hpp) for tools/trace_report.py."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])           # the experiment build instead of the product library
from yume_amd import vae_ops as V  # noqa: E402

out_dir = sys.argv[2]
os.makedirs(out_dir, exist_ok=True)
lib = _lib.load()
rd = lib.yume_debug_trace_read
rd.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
DEV = "cuda"
zero = torch.zeros(64, dtype=torch.bfloat16, device=DEV)
shapes = [("c256", 256, 256, 4, 352, 640, (3, 3, 3), False), ("c512", 512, 512, 4, 176, 320, (3, 3, 3), False),
          ("c1024", 1024, 1024, 2, 88, 160, (3, 3, 3), False), ("c1024_512", 1024, 512, 4, 176, 320, (3, 3, 3), False),
          ("up1024", 1024, 1024, 2, 88, 160, (1, 3, 3), True), ("c256_8f", 256, 256, 8, 352, 640, (3, 3, 3), False)]
for name, ci, co, T, H, W, k, ups in shapes:
    x = (torch.randn(T, H, W, ci, device=DEV) * 0.5).to(torch.bfloat16)
    cache = (torch.randn(2, H, W, ci, device=DEV) * 0.5).to(torch.bfloat16) if k[0] == 3 else None
    K = k[0] * k[1] * k[2] * ci
    w = (torch.randn(co, K, device=DEV) * K ** -0.5).to(torch.bfloat16)
    b = torch.zeros(co, device=DEV)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    out = torch.empty(T, Ho, Wo, co, dtype=torch.bfloat16, device=DEV)
    for _ in range(3):
        V.conv3d_cl(x, cache, w, b, co, k, (1, 1, 1), (k[0] - 1, 1, 1), ups, out, V.EPI_BF16, zero_page=zero)
    torch.cuda.synchronize()
    buf = np.zeros(32768 * 8, dtype=np.uint64)
    assert rd(buf.ctypes.data, buf.nbytes) == 0
    buf.tofile(os.path.join(out_dir, f"conv_{name}.bin"))
    print(name, "tiles", T * Ho * Wo // 256 * (co // 256), flush=True)
