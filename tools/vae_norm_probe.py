"""Times yume_vae_rmsnorm_silu at the row counts / channel widths of the two VAEs' full-size passes (HBM roofline: 2 B read + 2 B written per
element). Run once per setting of YUME_VAE_NORM_RI (read once per process): tools/vae_norm_probe.py > gpurun_out/...
"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yume_amd import vae_ops as V  # noqa: E402

SHAPES = [(16711680, 96), (8355840, 96), (4177920, 192), (2088960, 192), (522240, 384), (261120, 384),
          (3604480, 160), (901120, 320), (225280, 640), (14417920, 160), (3604480, 320)]


def main():
    dev = torch.device("cuda:0")
    print("YUME_VAE_NORM_RI =", os.environ.get("YUME_VAE_NORM_RI", "(default)"))
    for M, C in SHAPES:
        if M * C * 4 > 24e9:
            continue
        x = torch.randn((M, C), device=dev, dtype=torch.float32).bfloat16().view(1, 1, M, C)
        g = torch.ones(C, device=dev)
        out = torch.empty_like(x)
        for _ in range(2):
            V.rmsnorm_silu(x, g, True, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            V.rmsnorm_silu(x, g, True, out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"M={M:9d} C={C:4d}  {ms:8.3f} ms  {4.0 * M * C / (ms * 1e-3) / 1e12:6.2f} TB/s  frac {4.0 * M * C / (ms * 1e-3) / 8e12:.3f}")
        del x, out


if __name__ == "__main__":
    main()
