#!/usr/bin/env python3
"""merge_pmc_gemm.py <dir of tools/run_pmc_gemm.sh> <out.csv>: the per-shape PMC summaries (<dir>/<shape>.csv) as ONE table with a `group` column
(gemm_<shape>), the format bench.py::pmc_traffic_bytes reads for the GEMM groups."""
import csv, glob, os, sys
rows, cols = [], ["group"]
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.csv"))):
    shape = os.path.basename(f)[:-4]
    for r in csv.DictReader(open(f)):
        r = dict(r)
        r["group"] = "gemm_" + shape
        rows.append(r)
        for c in r:
            if c not in cols:
                cols.append(c)
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=cols)
    w.writeheader()
    w.writerows(rows)
print(len(rows), "rows ->", sys.argv[2])
