#!/usr/bin/env python3
"""trace8.py <trace.bin>: the stamps a -DYUME_TRACE build of attn_fwd8.hip leaves (tools/build_variant.sh trace8 attn_fwd8.hip -DYUME_TRACE;
tools/attn_check --lib yume_amd/lib/exp/libyume_hip_trace8.so --one Lq Lk H --trace f.bin 776). A workgroup overwrites its stamps at every item, so the
file holds each workgroup's LAST item boundary: 3 = in front of the tile before the item's last, 4 = behind it (bubble 1 begins), 5 = Q' has
landed, 6 = behind the boundary tile, 1 = range vote done, 2 = O^T stored, next item begins. s_memrealtime ticks are 10 ns."""
import struct
import sys

data = open(sys.argv[1], "rb").read()
w = struct.unpack(f"<{len(data) // 8}Q", data)
rows = []
for b in range(len(w) // 8):
    s = w[b * 8:b * 8 + 8]
    if s[0] == 0 or s[2] == 0 or s[6] == 0:
        continue
    rows.append(s)
# (stamp 2 = the top of an item that is not the workgroup's first: 2 -> 3 is that item's tiles up to the one before its last)
print(len(rows), "workgroups with a complete item boundary")
names = [("item start -> tile before last", 2, 3), ("tile before last", 3, 4), ("bubble 1 (Q' + ticket)", 4, 5), ("boundary tile", 5, 6), ("range vote", 6, 1), ("O^T store + drain", 1, 2)]
for name, a, b in names:
    d = sorted((r[b] - r[a]) * 0.01 for r in rows if r[b] > r[a])
    if d:
        print(f"{name:28s} n={len(d):4d}  min {d[0]:6.2f}  p50 {d[len(d) // 2]:6.2f}  max {d[-1]:6.2f} us")
