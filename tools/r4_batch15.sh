#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b15; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_fused_gpu.py tests/test_decoder_gpu.py -x -q -m gpu -k "conv or temporal or composed or fpn or writer or level" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log
AB_ARGS="--in-flight 1" tools/ab_bench.sh r4b15_ab default build/lib_conv_generic.so default build/lib_conv_generic.so
AB_ARGS="" tools/ab_bench.sh r4b15_ab4 default build/lib_conv_generic.so
