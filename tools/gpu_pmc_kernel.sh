#!/bin/bash
# usage (GPU box): tools/gpu_pmc_kernel.sh <tag> <kernel substring> "<counters pass 1>" ["<counters pass 2>" ...]
# one rocprofv3 --pmc pass per counter group over a short bench run; prints the per-launch averages of the named kernel
tag=$1; ksub=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-stress --steps 3 --warmup 2 > $out/p$i.json 2> $out/p$i.log || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
python3 - "$out" "$ksub" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out, ksub = sys.argv[1:3]
acc = defaultdict(list)
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[len(v) // 3:]
    print(f"{k:32s} {sum(v) / len(v):16.1f}  ({len(v)} launches)")
PY
