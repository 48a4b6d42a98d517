#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b13; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_fused_gpu.py -x -q -m gpu -k "outproj or split_precision or mixing or generator or temporal or conv" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
AB_ARGS="--in-flight 1" tools/ab_bench.sh r4b13_ab default build/lib_gs_skip.so default build/lib_gs_skip.so
AB_ARGS="" tools/ab_bench.sh r4b13_ab4 default build/lib_gs_skip.so
