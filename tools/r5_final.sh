#!/bin/bash
# round-end rehearsal (round 5) in the driver's order: build() (a no-op on the box: the built library travels with the snapshot), the whole -m gpu
# suite, smoke(), the default bench.  (Profiles: tools/gpu_final_profiles.sh <tag> pmc, then ... bench.)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5fin_reh; mkdir -p $out
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1; echo "build rc=$?"; tail -1 $out/build.log
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -2 $out/gpu_tests.log
[ $rc = 0 ] || exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
timeout -k 10 500 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('$out/bench_default.json')); print('value', round(d['value'],1), 'one', round(d['one_sample_in_flight']['value'],1), 'lanes', d['lanes_match_single_plan_bitwise'], 'frac', round(d['roofline']['frac'],3), 'cpu', round(d['cpu_baseline']['value'],3))"
timeout -k 10 500 python3 bench.py --config f8_3cam > $out/bench_f8_3cam.json 2> $out/bench_f8_3cam.err; echo "bench 3cam rc=$?"
