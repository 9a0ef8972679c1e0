"""Times yume_vae_dupup_add / yume_vae_avgdown_add at shapes of the Wan2.2 VAE's full-size passes (HBM roofline: dupup 2 + 2 bytes per output
element + the small input once; avgdown the input once + 2 + 2 bytes per output element). YUME_VAE_SHORTCUT_LINES=0 selects the general kernels."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yume_amd import vae_ops as V  # noqa: E402


def timed(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    print("YUME_VAE_SHORTCUT_LINES =", os.environ.get("YUME_VAE_SHORTCUT_LINES", "(default)"))
    for (Tin, Hin, Win, Cin, Cout, ft) in [(4, 352, 640, 512, 512, 2), (4, 176, 320, 1024, 512, 2), (8, 352, 640, 512, 256, 1), (2, 88, 160, 1024, 1024, 2)]:
        x = torch.randn((Tin, Hin, Win, Cin), device=dev).bfloat16()
        y = torch.randn((Tin * ft, Hin * 2, Win * 2, Cout), device=dev).bfloat16()
        ms = timed(lambda: V.dupup_add(x, y, ft, 2, 0))
        nbytes = 4.0 * y.numel() + 2.0 * x.numel()
        print(f"dupup   x {tuple(x.shape)} -> y {tuple(y.shape)}: {ms:7.3f} ms  {nbytes / (ms * 1e-3) / 1e12:5.2f} TB/s  frac {nbytes / (ms * 1e-3) / 8e12:.3f}")
        del x, y
    for (Tin, Hin, Win, Cin, Cout, ft) in [(4, 704, 1280, 160, 160, 1), (4, 352, 640, 160, 320, 2), (2, 176, 320, 320, 640, 2)]:
        x = torch.randn((Tin, Hin, Win, Cin), device=dev).bfloat16()
        y = torch.randn((Tin // ft, Hin // 2, Win // 2, Cout), device=dev).bfloat16()
        ms = timed(lambda: V.avgdown_add(x, y, ft, 2))
        nbytes = 4.0 * y.numel() + 2.0 * x.numel()
        print(f"avgdown x {tuple(x.shape)} -> y {tuple(y.shape)}: {ms:7.3f} ms  {nbytes / (ms * 1e-3) / 1e12:5.2f} TB/s  frac {nbytes / (ms * 1e-3) / 8e12:.3f}")
        del x, y


if __name__ == "__main__":
    main()
