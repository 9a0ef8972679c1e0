"""The row split's remainder launches of the 5B block (gemm_bf16.hip: ffn.0 244 rows x 14336, QKV 500 rows x 9216, K = 3072) on the 128x128 kernel
(variant 1), on the one-wave-per-SIMD kernel (variant 3) and, where the epilogue allows, through yume_gemm_bf16_splitk with 2 / 4 slices."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yume_amd import _lib, ops  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    for (M, N, K, epi, name) in [(244, 14336, 3072, ops.EPI_BF16_GELU, "ffn.0 remainder"), (500, 9216, 3072, ops.EPI_BF16, "QKV remainder (plain bf16 epilogue)"),
                                 (244, 3072, 14336, ops.EPI_F32, "ffn.2-like 244 rows")]:
        a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
        w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
        b = torch.randn(N, device=dev) * 0.1
        out = torch.empty((M, N), dtype=torch.float32 if epi == ops.EPI_F32 else torch.bfloat16, device=dev)
        res = {}
        for v in (1, 2, 3):
            try:
                res[f"variant {v}"] = timed(lambda: ops.gemm_bf16(a, w, b, out, epi, variant=v))
            except RuntimeError as e:
                res[f"variant {v}"] = str(e)[:40]
        ref = out.clone()
        for splits in (2, 4):
            ws = torch.empty(splits * M * N, dtype=torch.float32, device=dev)

            def go():
                rc = lib.yume_gemm_bf16_splitk(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), M, N, K, epi, out.data_ptr(), N, splits, ws.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream)
                _lib.check(rc, "splitk")
            res[f"splitk {splits}"] = timed(go)
            res[f"splitk {splits} maxdiff"] = float((out.float() - ref.float()).abs().max())
        print(name, f"M={M} N={N} K={K}:", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()})


if __name__ == "__main__":
    main()
