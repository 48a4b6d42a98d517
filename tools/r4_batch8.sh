#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b8; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -6 $out/gpu_tests.log
