"""trace_report.py <trace.bin> [--blocks N]: reads the per-workgroup stamps of a -DYUME_TRACE experiment build (csrc/trace.hpp; written by
tools/attn_check --trace / tools/gemm_check --trace) and says where a launch's time went: the launch's span, how long workgroups spent in
their K loop and in their epilogue, the gaps a CU stood empty between two workgroups, the idle tail per CU and per XCD.
Times: s_memrealtime ticks of 10 ns."""
import sys
import numpy as np


def main():
    t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
    live = t[:, 0] > 0
    if "--blocks" in sys.argv:
        n = int(sys.argv[sys.argv.index("--blocks") + 1])
        live[n:] = False
    t = t[live]
    t0, t1, t2 = (t[:, i].astype(np.int64) for i in range(3))
    done = t2 >= t0                       # blocks that returned before the last stamp (surplus ids) leave a stale t2
    xcc = (t[:, 7] >> np.uint64(32)).astype(np.int64) & 0xf
    hw = (t[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
    cu = xcc * 4096 + ((hw >> 8) & 0xff)  # (XCC, SE | SH | CU)
    beg, end = t0.min(), t2[done].max()
    us = lambda x: x / 100.0
    print(f"{len(t)} workgroups, {len(np.unique(cu))} CUs, span {us(end - beg):.1f} us")
    loop, epi = (t1 - t0)[done], (t2 - t1)[done]
    print(f"K loop  per workgroup: mean {us(loop.mean()):.2f}  min {us(loop.min()):.2f}  p50 {us(np.median(loop)):.2f}  max {us(loop.max()):.2f} us")
    print(f"epilogue per workgroup: mean {us(epi.mean()):.2f}  min {us(epi.min()):.2f}  p50 {us(np.median(epi)):.2f}  max {us(epi.max()):.2f} us")
    for s in range(3, 7):           # finer points, relative to the workgroup's entry
        v = t[:, s].astype(np.int64)
        m = done & (v >= t0)
        if m.sum():
            print(f"  point {s}: {m.sum()} workgroups, +{us((v - t0)[m].mean()):.2f} us after entry (min {us((v - t0)[m].min()):.2f}, max {us((v - t0)[m].max()):.2f}); "
                  f"{us((t2 - v)[m].mean()):.2f} us before the end")
    # segments between successive points (K loop done = 1, then 3, 4, 5, 6 where the kernel set them, then the end = 2)
    order = [1] + [k for k in range(3, 7) if (done & (t[:, k].astype(np.int64) >= t0)).sum() == done.sum()] + [2]
    if len(order) > 2:
        for a, b in zip(order[:-1], order[1:]):
            d = (t[:, b].astype(np.int64) - t[:, a].astype(np.int64))[done]
            print(f"  segment {a} -> {b}: mean {us(d.mean()):.2f}  p10 {us(np.percentile(d, 10)):.2f}  p50 {us(np.median(d)):.2f}  p90 {us(np.percentile(d, 90)):.2f}  max {us(d.max()):.2f} us")
    busy_tot, gap_tot, tail_tot, head_tot, ngap = 0, 0, 0, 0, 0
    per_x = {}
    for c in np.unique(cu):
        m = (cu == c) & done
        o = np.argsort(t0[m])
        a, b = t0[m][o], t2[m][o]
        busy = (b - a).sum()
        gaps = (a[1:] - b[:-1])
        busy_tot += busy
        gap_tot += gaps.clip(min=0).sum()
        ngap += len(gaps)
        tail_tot += end - b[-1]
        head_tot += a[0] - beg
        x = c // 4096
        d = per_x.setdefault(x, dict(n=0, wg=0, last=0, first=1 << 62, busy=0))
        d["n"] += 1; d["wg"] += len(a); d["last"] = max(d["last"], b[-1]); d["first"] = min(d["first"], a[0]); d["busy"] += busy
    ncu = len(np.unique(cu))
    span = end - beg
    print(f"per CU, of the span: resident {100 * busy_tot / ncu / span:.1f} %  gaps between workgroups {100 * gap_tot / ncu / span:.1f} % "
          f"({us(gap_tot / max(ngap, 1)):.2f} us each, {ngap / ncu:.1f} per CU)  idle tail {100 * tail_tot / ncu / span:.1f} %  head {100 * head_tot / ncu / span:.1f} %")
    print(f"of the resident time: K loop {100 * loop.sum() / (loop.sum() + epi.sum()):.1f} %  epilogue {100 * epi.sum() / (loop.sum() + epi.sum()):.1f} %")
    for x in sorted(per_x):
        d = per_x[x]
        print(f"  XCD {x}: {d['n']} CUs  {d['wg']} workgroups  first start +{us(d['first'] - beg):.1f} us  last end {us(end - d['last']):.1f} us before the end"
              f"  resident {100 * d['busy'] / d['n'] / span:.1f} %")
    # dispatch order: is the block id order the time order?
    ids = np.nonzero(live)[0]
    first_round = np.sort(t0)[: ncu]
    print(f"first round of starts spread over {us(first_round[-1] - first_round[0]):.2f} us")
    # durations by round (k-th workgroup of each CU)
    rounds = {}
    for c in np.unique(cu):
        m = (cu == c) & done
        o = np.argsort(t0[m])
        for k, (a, b1, b2) in enumerate(zip(t0[m][o], t1[m][o], t2[m][o])):
            rounds.setdefault(k, []).append((b1 - a, b2 - b1, a - beg))
    for k in sorted(rounds):
        r = np.array(rounds[k])
        print(f"  round {k}: {len(r)} workgroups  start +{us(r[:, 2].mean()):.1f} us (spread {us(r[:, 2].max() - r[:, 2].min()):.1f})  loop {us(r[:, 0].mean()):.2f} us  epilogue {us(r[:, 1].mean()):.2f} us")


if __name__ == "__main__":
    main()
