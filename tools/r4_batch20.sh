#!/bin/bash
# fp32 value streams as the default again, int16 block storage (producer epilogues) opt-in: whole -m gpu suite, smoke, default bench
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b20; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -6 $out/gpu_tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout -k 10 500 python3 bench.py --no-stress > $out/bench_f8.json 2> $out/bench_f8.err; echo "bench f8 rc=$?"; tail -3 $out/bench_f8.err
python3 - <<PY
import json
d=json.load(open("$out/bench_f8.json"))
print("headline", d["value"], "one", d["one_sample_in_flight"]["value"], "lanes ok", d["lanes_match_single_plan_bitwise"])
print("i16 line", json.dumps({k:v for k,v in d.get("bev_values_int16_block",{}).items() if k!="note"}))
PY
