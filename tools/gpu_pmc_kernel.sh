#!/bin/bash
# usage (GPU box): tools/gpu_pmc_kernel.sh <tag> <kernel substring> "<counters pass 1>" ["<counters pass 2>" ...]
# one rocprofv3 --pmc pass per counter group over a short bench run; prints the per-launch averages of the named kernel
# PMC_ARGS: extra bench.py arguments (default "--no-graph --in-flight 1": eager launches, counters serialise the kernels anyway)
# PMC_JSON: if set, the averages are also written there as JSON (what profiles/r04_pmc_bev.json is made from)
tag=$1; ksub=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
failed=""
for grp in "$@"; do
  i=$((i+1))
  rm -rf $out/p$i $out/p$i.json $out/p$i.log      # no stale counter CSVs of an earlier run under the same tag
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-stress --steps 3 --warmup 2 ${PMC_ARGS:---no-graph --in-flight 1} > $out/p$i.json 2> $out/p$i.log || { echo "pass $i ($grp) failed"; tail -3 $out/p$i.log; failed="$failed|$grp"; rm -rf $out/p$i; }
done
FAILED_GROUPS="$failed" python3 - "$out" "$ksub" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out, ksub = sys.argv[1:3]
acc = defaultdict(list)
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in sorted(acc.items()):
    v = v[len(v) // 3:]
    res[k] = {"avg_per_launch": sum(v) / len(v), "launches": len(v)}
    print(f"{k:32s} {sum(v) / len(v):16.1f}  ({len(v)} launches)")
if os.environ.get("PMC_JSON"):
    import json
    json.dump({"kernel_substring": ksub, "counters": res,
               "failed_groups": [g for g in os.environ.get("FAILED_GROUPS", "").split("|") if g]},
              open(os.environ["PMC_JSON"], "w"), indent=1)
PY
# a failed pass is recorded in the JSON and makes the script fail: a profile that lacks a counter group must not look complete
[ -z "$failed" ] || { echo "failed counter groups:$failed"; exit 1; }
