#!/usr/bin/env python3
"""Per-kernel averages of every counter found under a directory of rocprofv3 --pmc passes (tools/gpu_pmc_kernel.sh output).
usage: pmc_all_kernels.py <dir> [out.json] [kernel substring ...]"""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2].endswith(".json") else None
subs = [a for a in sys.argv[2:] if not a.endswith(".json")]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if subs and not any(s in k for s in subs):
            continue
        acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k in sorted(acc):
    res[k] = {}
    print(k)
    for c, v in sorted(acc[k].items()):
        v = v[len(v) // 3:]
        res[k][c] = {"avg_per_launch": sum(v) / len(v), "launches": len(v)}
        print(f"    {c:34s} {sum(v) / len(v):18.1f}  ({len(v)})")
if out:
    json.dump(res, open(out, "w"), indent=1)
