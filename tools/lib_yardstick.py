#!/usr/bin/env python3
"""Yardstick only (not a product path): what the ROCm libraries behind torch (hipBLASLt / rocBLAS GEMM, the SDPA backends) reach on the
block's shapes on this box, next to the hand-written kernels in the same call. Random bf16 data. Writes gpurun_out/yardstick.json."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, warm=3, iters=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    res = {}
    bf = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)
    shapes = {"qkv": (9460, 9216, 3072), "o": (9460, 3072, 3072), "ffn0": (9460, 14336, 3072), "ffn2": (9460, 3072, 14336),
              "sq8192": (8192, 8192, 8192), "14b_ffn0": (27810, 13824, 5120)}
    for name, (M, N, K) in shapes.items():
        a, w, b = bf(M, K), bf(N, K) * (K ** -0.5), torch.randn(N, device=DEV)
        o = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        bb = b.bfloat16()
        t_lib = timeit(lambda: torch.nn.functional.linear(a, w, bb))
        t_own = timeit(lambda: ops.gemm_bf16(a, w, b, o, ops.EPI_BF16))
        fl = 2 * M * N * K
        res[f"gemm_{name}"] = {"lib_ms": t_lib, "lib_tflops": fl / t_lib / 1e9, "own_ms": t_own, "own_tflops": fl / t_own / 1e9}
        print(f"gemm {name} {M}x{N}x{K}: library {t_lib:.3f} ms {fl/t_lib/1e9:.0f} TF | own {t_own:.3f} ms {fl/t_own/1e9:.0f} TF", flush=True)
        del a, w, o
    for (L, H, tag) in ((9460, 24, "5b"), (23460, 40, "14b")):
        q, k, v = bf(1, H, L, 128), bf(1, H, L, 128), bf(1, H, L, 128)
        fl = 4 * L * L * 128 * H
        for backend in ("flash", "efficient", "default"):
            try:
                from torch.nn.attention import SDPBackend, sdpa_kernel
                ctx = {"flash": [SDPBackend.FLASH_ATTENTION], "efficient": [SDPBackend.EFFICIENT_ATTENTION]}.get(backend)
                if ctx:
                    with sdpa_kernel(ctx):
                        t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v), warm=2, iters=4)
                else:
                    t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v), warm=2, iters=4)
                res[f"sdpa_{tag}_{backend}"] = {"ms": t, "tflops": fl / t / 1e9}
                print(f"sdpa {tag} {backend}: {t:.3f} ms {fl/t/1e9:.0f} TF", flush=True)
            except Exception as ex:  # a backend that is not built into this wheel
                print(f"sdpa {tag} {backend}: unavailable ({type(ex).__name__}: {str(ex)[:120]})", flush=True)
        qq = q[0].permute(1, 0, 2).reshape(L, H * 128).contiguous()
        kk = k[0].permute(1, 0, 2).reshape(L, H * 128).contiguous()
        vt = torch.empty(H * 128, (L + 7) // 8 * 8, dtype=torch.bfloat16, device=DEV)
        ops.transpose_bf16(v[0].permute(1, 0, 2).reshape(L, H * 128).contiguous(), vt)
        o = torch.empty(L, H * 128, dtype=torch.bfloat16, device=DEV)
        t = timeit(lambda: ops.attn_fwd(qq, kk, vt, o, L, L, H), warm=2, iters=4)
        res[f"attn_{tag}_own"] = {"ms": t, "tflops": fl / t / 1e9}
        print(f"attention {tag} own: {t:.3f} ms {fl/t/1e9:.0f} TF", flush=True)
        del q, k, v, qq, kk, vt, o
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "yardstick.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
