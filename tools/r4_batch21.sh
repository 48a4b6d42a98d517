#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b21; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu --deselect tests/test_parity_gpu.py --deselect tests/test_fused_gpu.py --deselect tests/test_ops_gpu.py --deselect tests/test_decoder_gpu.py -s > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -6 $out/gpu_tests.log; grep "head_small6 box error" $out/gpu_tests.log
