#!/bin/bash
# round-4 final validation, part 1: the whole -m gpu suite, smoke(), then the profiles (kernel stats with four plans / one plan in flight,
# PMC traffic passes for both configs) -- summarise with tools/pmc_traffic.py, then run part 2 (tools/gpu_final_profiles.sh <tag> bench)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4fin; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -4 $out/gpu_tests.log
[ $rc = 0 ] || exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
tools/gpu_final_profiles.sh r4fin pmc
