#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one dir per pass) into per-kernel averages. usage: pmc_summary.py <pmc_dir> [out.csv]"""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:110]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(root, "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:110]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k in acc:
    if not any(t in k for t in ("gemm", "attn", "adaln", "conv", "Cijk", "norm")): continue
    d = {"kernel": k, "avg_us(profiled)": sum(dur[k]) / max(1, len(dur[k]))}
    for c, v in acc[k].items():
        d[c] = sum(v) / len(v)
    rows.append(d)
cols = ["kernel", "avg_us(profiled)"] + sorted({c for r in rows for c in r if c not in ("kernel", "avg_us(profiled)")})
for r in rows:
    print(r["kernel"])
    for c in cols[1:]:
        if c in r: print(f"   {c:28s} {r[c]:16.1f}")
if len(sys.argv) > 2:
    with open(sys.argv[2], "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=cols); w.writeheader(); w.writerows(rows)
