#!/usr/bin/env python3
"""isa_blocks.py <file.s> <kernel name substring>: per basic block of a kernel's gfx950 assembly — instructions, MFMAs, scratch loads / stores and
the vmcnt values waited for. A quick map of where hipcc parked spill traffic and memory waits relative to the MFMA-dense blocks."""
import re
import sys

txt = open(sys.argv[1]).read()
body = txt[txt.index(sys.argv[2]):]
end = body.find(".Lfunc_end")
lines = body[:end if end > 0 else None].split("\n")
blocks, cur = [], ["entry", 0, []]
for i, l in enumerate(lines):
    if l.startswith(".LBB"):
        blocks.append(cur)
        cur = [l.split(":")[0], i, []]
    elif l.strip() and l.strip()[0] not in ";.":
        cur[2].append(l.strip())
blocks.append(cur)
pat = re.compile(r"vmcnt\((\d+)\)")
for name, start, ins in blocks:
    nm = sum("v_mfma" in x for x in ins)
    sl = sum("scratch_load" in x for x in ins)
    ss = sum("scratch_store" in x for x in ins)
    vm = [pat.search(x).group(1) for x in ins if "vmcnt(" in x]
    if nm >= 16 or sl or ss:
        print(f"{name:12s} @{start:6d} n={len(ins):5d} mfma={nm:4d} scratch ld/st={sl}/{ss} vmcnt={vm}")
print("total", sum(len(b[2]) for b in blocks), "instructions,", sum(sum("scratch_" in x for x in b[2]) for b in blocks), "scratch ops")
