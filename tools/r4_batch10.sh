#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b10; mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_graph_gpu.py -x -q -m gpu -k "regroup_beside" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log
for v in "" "--overlap-regroup" "" "--overlap-regroup"; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-stress --steps 30 $v > $out/b.json 2> $out/b.err || { echo "bench $v FAILED"; tail -3 $out/b.err; }
  python3 - "$v" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r4b10/b.json"))
print("%-18s value %.1f one %.1f lanes %s" % (sys.argv[1] or "serial", d["value"], d["one_sample_in_flight"]["value"], d["lanes_match_single_plan_bitwise"]))
PY
done
for n in 3 5 6; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-stress --steps 30 --in-flight $n > $out/b.json 2> $out/b.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r4b10/b.json')); print('in-flight $n value %.1f'%d['value'], d['lanes_match_single_plan_bitwise'])"
done
