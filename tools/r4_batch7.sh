#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b7; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_fused_gpu.py tests/test_lowprec_storage_gpu.py tests/test_full_size_f2f4_gpu.py tests/test_capi_symbols.py -x -q -m gpu -s > $out/tests.log 2>&1; echo "tests rc=$?"; tail -5 $out/tests.log; grep -h "rac_.*f8\|values\|pyramid" $out/tests.log | cut -c1-250 | head -40
timeout -k 10 600 python3 bench.py > $out/bench_f8.json 2> $out/bench_f8.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4b7/bench_f8.json"))
print("value", d["value"], "one", d["one_sample_in_flight"], "lanes", d["lanes_match_single_plan_bitwise"])
print("i16", d.get("bev_values_int16_block"))
print("pregrouped", d.get("pregrouped_producer_layout"))
for k,v in d["roofline_stress"].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!="note" and a!="set"})
print(d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["bev_sampling"])
print(d["cpu_baseline"]["value"], d["parity_vs_oracle"])
PY
