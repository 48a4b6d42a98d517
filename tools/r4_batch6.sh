#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b6; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_fused_gpu.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log
AB_ARGS="--in-flight 1" tools/ab_bench.sh r4b6_ab default
AB_ARGS="" tools/ab_bench.sh r4b6_ab4 default
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof1 -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-stress --in-flight 1 --steps 10 > $GRAFT_REPO_ROOT/$out/bench_prof1.json 2> $GRAFT_REPO_ROOT/$out/prof1.log; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f=glob.glob("gpurun_out/r4b6/prof1/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:24]: print("%-60s %6s %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
