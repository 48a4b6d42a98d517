#!/bin/bash
# q16 epilogues of the two value-stream producers: op tests, decoder tests, timing of the i16 mode against fp32
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b18; mkdir -p $out
timeout -k 10 700 python3 -m pytest tests/test_fused_gpu.py tests/test_lowprec_storage_gpu.py tests/test_capi_symbols.py -x -q -m gpu -k "q16 or int16 or i16 or conv3x3 or value_proj or composed or temporal or capi" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -5 $out/tests.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-stress --steps 30 --in-flight 1 > $out/bench1.json 2> $out/bench1.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.load(open("$out/bench1.json"))
print("f32 one-plan", d["value"], "i16 line:", json.dumps(d.get("bev_values_int16_block")))
PY
