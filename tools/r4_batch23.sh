#!/bin/bash
# the adaptive sampling inside the mixing kernel: op test, then decoder-level tests, then a short bench A/B (fused default vs RAC_NO_FUSE)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b23; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_fused_gpu.py -x -q -m gpu -k "mixing or sampling4d" > $out/tests_op.log 2>&1; echo "op tests rc=$?"; tail -5 $out/tests_op.log
