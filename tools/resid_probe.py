#!/usr/bin/env python3
"""Timing of the residual-epilogue GEMMs of the 5B block (o, cross-o, ffn.2; gate*y + x in place on the fp32 residual stream) and of the
bf16-epilogue ones next to them, back-to-back launches. A/B of library builds: YUME_HIP_LIB=yume_amd/lib/exp/libyume_hip_<tag>.so python tools/resid_probe.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yume_amd import ops  # noqa: E402

DEV = "cuda"
L, C, FF = 9460, 3072, 14336


def timeit(fn, reps=100):
    for _ in range(30):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    g = torch.Generator(device=DEV).manual_seed(2)
    tab = torch.randn((2, 6, C), generator=g, device=DEV) * 0.1
    seg = (torch.arange(L, device=DEV) >= L // 7).to(torch.int32)          # two timestep segments (the FramePack path)
    alt = (torch.arange(L, device=DEV) % 2).to(torch.int32)                # interleaved rows: the two-deep per-row path
    x = torch.randn((L, C), generator=g, device=DEV)
    res = {"lib": os.environ.get("YUME_HIP_LIB", "product")}
    for name, K in (("o", C), ("ffn2", FF)):
        a = (torch.randn((L, K), generator=g, device=DEV) * 0.5).to(torch.bfloat16)
        w = (torch.randn((C, K), generator=g, device=DEV) * K ** -0.5 * 0.1).to(torch.bfloat16)
        b = torch.randn(C, generator=g, device=DEV) * 0.01
        res[name + "_gate_segments"] = round(timeit(lambda: ops.gemm_bf16(a, w, b, x, ops.EPI_RESID, gate=tab[:, 2], gate_stride=6 * C, row_idx=seg)), 2)
        if name == "o":
            res["o_gate_interleaved"] = round(timeit(lambda: ops.gemm_bf16(a, w, b, x, ops.EPI_RESID, gate=tab[:, 2], gate_stride=6 * C, row_idx=alt)), 2)
            res["cross_o_no_gate"] = round(timeit(lambda: ops.gemm_bf16(a, w, b, x, ops.EPI_RESID)), 2)
            res["o_gate_no_index"] = round(timeit(lambda: ops.gemm_bf16(a, w, b, x, ops.EPI_RESID, gate=tab[:, 2], gate_stride=6 * C)), 2)
            one = torch.ones(L, dtype=torch.int32, device=DEV)
            res["o_gate_index_all_one"] = round(timeit(lambda: ops.gemm_bf16(a, w, b, x, ops.EPI_RESID, gate=tab[:, 2], gate_stride=6 * C, row_idx=one)), 2)
            res["o_gate_segments_again"] = round(timeit(lambda: ops.gemm_bf16(a, w, b, x, ops.EPI_RESID, gate=tab[:, 2], gate_stride=6 * C, row_idx=seg)), 2)
            res["o_gate_interleaved_again"] = round(timeit(lambda: ops.gemm_bf16(a, w, b, x, ops.EPI_RESID, gate=tab[:, 2], gate_stride=6 * C, row_idx=alt)), 2)
            out = torch.empty((L, C), dtype=torch.bfloat16, device=DEV)
            res["cross_q_bf16"] = round(timeit(lambda: ops.gemm_bf16(a, w, b, out, ops.EPI_BF16)), 2)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
