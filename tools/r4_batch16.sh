#!/bin/bash
# generator with two four-wave workgroups per CU (default) against one eight-wave workgroup (build/lib_gen8.so)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b16; mkdir -p $out
timeout -k 10 500 python3 -m pytest tests/test_fused_gpu.py -x -q -m gpu -k "generator or mixing or outproj" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log
AB_ARGS="--in-flight 1" tools/ab_bench.sh r4b16_ab1 default build/lib_gen8.so default build/lib_gen8.so
AB_ARGS="" tools/ab_bench.sh r4b16_ab4 default build/lib_gen8.so default build/lib_gen8.so
