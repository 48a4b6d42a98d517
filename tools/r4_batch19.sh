#!/bin/bash
# int16 block value streams as the default storage: whole -m gpu suite, then the default bench (headline i16, fp32 comparison line)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b19; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -6 $out/gpu_tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout -k 10 500 python3 bench.py --no-stress > $out/bench_f8.json 2> $out/bench_f8.err; echo "bench f8 rc=$?"; tail -3 $out/bench_f8.err
python3 - <<PY
import json
d=json.load(open("$out/bench_f8.json"))
print("headline", d["value"], "one", d["one_sample_in_flight"]["value"], "lanes ok", d["lanes_match_single_plan_bitwise"])
print("fp32 line", json.dumps(d.get("fp32_value_streams")))
print("parity", json.dumps(d.get("parity_vs_oracle"))[:600])
print("bev", json.dumps(d["roofline"]["bev_sampling"]))
PY
