#!/usr/bin/env python3
"""Print per-kernel register/LDS/occupancy usage of csrc/*.hip (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os, glob
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "yume_amd/csrc/*.hip")))
for f in files:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-DNDEBUG",
                        "-I", os.path.join(root, "include"), "-c", f, "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m: 
            if "error" in line: print(line)
            continue
        s = m.group(1).strip()
        if s.startswith("Function Name:"):
            cur = {"name": s.split(":",1)[1].strip()}
        elif ":" in s:
            k, v = s.split(":", 1); cur[k.strip()] = v.strip()
            if k.strip().startswith("LDS Size"):
                name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
                print(f"{os.path.basename(f):18s} {name[:70]:70s} VGPR {cur.get('VGPRs','?'):>4} AGPR {cur.get('AGPRs','?'):>3} SGPR {cur.get('TotalSGPRs','?'):>3} spill {cur.get('VGPRs Spill','?')} scratch {cur.get('ScratchSize [bytes/lane]','?')} occ {cur.get('Occupancy [waves/SIMD]','?')} LDS {cur.get('LDS Size [bytes/block]','?')}")
