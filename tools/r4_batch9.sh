#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b9; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_fused_gpu.py tests/test_lowprec_storage_gpu.py -x -q -m gpu -k "int16 or bev_sampling" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-stress > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4b9/bench.json"))
print("value", d["value"], "one", d["one_sample_in_flight"]["value"], d["lanes_match_single_plan_bitwise"])
print("i16", {k:v for k,v in d["bev_values_int16_block"].items() if k!="note"})
PY
