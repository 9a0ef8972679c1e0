"""tools/trace_report.py on a synthetic trace (the layout csrc/trace.hpp writes: 8 words per workgroup — entry, K loop done, end, four finer
points, XCC_ID << 32 | HW_ID — in 10 ns ticks): two rounds of workgroups on 16 CUs of 2 XCDs, one XCD slower than the other."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT


def test_trace_report_reads_the_layout_and_finds_the_idle_tail(tmp_path):
    n_cu, rounds = 16, 2
    t = np.zeros((32768, 8), dtype=np.uint64)
    base = 1_000_000
    wg = 0
    for r in range(rounds):
        for cu in range(n_cu):
            xcc = cu // 8
            speed = 1.0 if xcc == 0 else 1.25                       # XCD 1 is 25 % slower
            start = base + int(r * 10_000 * speed) + 5 * cu         # 100 us per workgroup on the fast XCD
            loop = int(9_000 * speed)
            epi = int(1_000 * speed)
            t[wg, 0], t[wg, 1], t[wg, 2] = start, start + loop, start + loop + epi
            t[wg, 3] = start + loop + epi // 2                      # a finer point inside the epilogue
            t[wg, 7] = (xcc << 32) | ((cu % 8) << 8)                # HW_ID: CU id in bits 8..11
            wg += 1
    f = tmp_path / "trace.bin"
    t.tofile(f)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_report.py"), str(f)], capture_output=True, text=True, check=True).stdout
    assert f"{n_cu * rounds} workgroups, {n_cu} CUs" in out
    assert "K loop  per workgroup: mean 101.25" in out              # (90 + 112.5) / 2 us
    assert "epilogue per workgroup: mean 11.25" in out
    assert "point 3: 32 workgroups" in out
    assert "XCD 0: 8 CUs  16 workgroups" in out and "XCD 1: 8 CUs  16 workgroups" in out
    line = next(l for l in out.split("\n") if l.startswith("  XCD 0"))
    assert "last end 50." in line                                   # the fast XCD is done 50 us before the slow one
    tail = float(out.split("idle tail ")[1].split(" %")[0])
    assert 9.0 < tail < 11.0                                        # half of the CUs idle for 50 of 250 us
