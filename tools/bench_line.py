"""bench_line.py: reads bench.py's JSON line from stdin and prints the step time and the per-group table in one short line each."""
import json
import sys

d = json.loads([l for l in sys.stdin.read().split("\n") if l.startswith("{")][-1])
print(f"{d['ms_per_step']:.2f} ms/step  {d['value']:.3f} {d['unit']}")
for g in d.get("roofline_all", []):
    print("   " + "  ".join(f"{k}={v:.4g}" if isinstance(v, float) else f"{k}={v}" for k, v in g.items() if k in ("group", "name", "ms_per_step", "launch_ms", "achieved", "frac")))
