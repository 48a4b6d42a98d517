#!/bin/bash
# usage (on the GPU box, through gpurun): tools/gpu_bench_trace.sh <tag> [bench args]
# plain bench run (JSON line) + a rocprofv3 kernel-trace run of the same command and its per-forward timeline
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 bench.py "$@" > $out/bench.json 2> $out/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-stress --steps 10 "$@" > $out/bench_prof.json 2> $out/prof.log || exit 1
cd $GRAFT_REPO_ROOT
python3 tools/trace_timeline.py $out/prof/run_results.db 3 > $out/timeline.txt
python3 - <<PY
import json
d=json.load(open("$out/bench.json"))
print("value", round(d["value"],2), "ms/step", round(d["ms_per_step"],3))
r=d["roofline"]; print("sampling4d ms", r["avg_launch_ms"], "frac", r["frac"], "bev", r["bev_sampling"]["avg_launch_ms"])
for k,v in d["mfma"].items(): print(k, round(v["avg_launch_ms"]*1e3,1),"us frac", round(v["frac"],3))
print(d.get("parity_vs_oracle"))
PY
grep -A 60 "^kernel time" $out/timeline.txt | head -45
