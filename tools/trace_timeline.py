#!/usr/bin/env python3
"""Timeline of ONE forward from a rocprofv3 --kernel-trace CSV: every dispatch in order with its duration and the idle gap
before it; plus per-kernel totals of that forward.  usage: trace_timeline.py <kernel_trace.csv | results.db> [forward index from the end, default 2]"""
import csv
import sys
from collections import defaultdict

if sys.argv[1].endswith(".db"):      # rocprofv3's default output (rocpd sqlite): the `kernels` view
    import sqlite3
    q = "select name, start, end, grid_x, grid_y, grid_z, vgpr_count, lds_size from kernels"
    rows = [dict(Kernel_Name=n, Start_Timestamp=s, End_Timestamp=e, Grid_Size_X=gx, Grid_Size_Y=gy, Grid_Size_Z=gz, VGPR_Count=v,
                 LDS_Block_Size=l) for n, s, e, gx, gy, gz, v, l in sqlite3.connect(sys.argv[1]).execute(q)]
else:
    rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "regroup" in r["Kernel_Name"]]
per_fwd = 1 if any("regroup_multi" in r["Kernel_Name"] for r in rows) else 4     # regroup launches per forward
starts = idx[::per_fwd]
a, b = starts[-back - 1], starts[-back]
t0 = prev = int(rows[a]["Start_Timestamp"])
tot = gap = 0
agg = defaultdict(lambda: [0, 0])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0][-70:]
    print(f"{(s - t0) / 1e3:8.1f} gap {(s - prev) / 1e3:6.1f}  {(e - s) / 1e3:7.1f} us  {n}  grid={r.get('Grid_Size_X', '')},{r.get('Grid_Size_Y', '')},{r.get('Grid_Size_Z', '')} vgpr={r.get('VGPR_Count','')} lds={r.get('LDS_Block_Size','')}")
    tot += e - s
    gap += max(0, s - prev)
    prev = max(prev, e)
    agg[n][0] += 1
    agg[n][1] += e - s
print(f"kernel time {tot / 1e3:.1f} us, idle gaps {gap / 1e3:.1f} us, wall {(prev - t0) / 1e3:.1f} us, {b - a} dispatches")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:4d} x {t / c / 1e3:8.1f} us = {t / 1e3:8.1f} us  {n}")
