#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b12; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_bev_pool.py tests/test_full_size_f2f4_gpu.py -x -q -m gpu -s > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log; grep -h "rac_bev_pool" $out/tests.log
timeout -k 10 300 python3 bench.py --stress-only > $out/stress.json 2> $out/stress.err; echo "stress rc=$?"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4b12/stress.json"))["roofline_stress"]
for k,v in d.items(): print("%-22s %.1f us frac %.3f %s"%(k, v["avg_launch_ms"]*1e3, v["frac"], ("atomic_frac %.3f"%v["atomic_frac"]) if "atomic_frac" in v else ""))
PY
