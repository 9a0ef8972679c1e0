#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) into a per-kernel stats table (CSV + text)."""
import sqlite3, sys, subprocess, csv
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
namecol = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[1])
rows = db.execute(f"select s.{namecol}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                  f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{namecol} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
out = sys.argv[2] if len(sys.argv) > 2 else None
lines = [("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct")]
for n, c, s, mn, mx in rows:
    lines.append((n[:110], c, f"{s/1e6:.3f}", f"{s/c/1e3:.2f}", f"{mn/1e3:.2f}", f"{mx/1e3:.2f}", f"{100*s/tot:.2f}"))
if out:
    with open(out, "w", newline="") as f:
        csv.writer(f).writerows(lines)
for l in lines[:25]:
    print("{:110s} {:>6} {:>10} {:>10} {:>9} {:>9} {:>6}".format(*map(str, l)))
print(f"total kernel time {tot/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
