#!/bin/bash
# round-4 final validation: the whole -m gpu suite, smoke(), then the default bench for both configs
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4fin; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $out/gpu_tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout -k 10 500 python3 bench.py > $out/bench_f8.json 2> $out/bench_f8.err; echo "bench f8 rc=$?"
timeout -k 10 500 python3 bench.py --config f8_3cam > $out/bench_f8_3cam.json 2> $out/bench_f8_3cam.err; echo "bench 3cam rc=$?"
