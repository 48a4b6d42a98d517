#!/bin/bash
# round-4 GPU batch 2: the whole -m gpu suite on the current tree, then the 16-bit storage measurements
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b2; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -5 $out/gpu_tests.log
timeout -k 10 600 python3 tools/exp_lowprec.py $out/exp_lowprec.json > $out/exp_lowprec.log 2>&1; echo "lowprec rc=$?"; cat $out/exp_lowprec.log | tail -40
