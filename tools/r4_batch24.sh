#!/bin/bash
# the adaptive sampling inside the mixing kernel: decoder-level parity tests, then the A/B against the two stand-alone kernels
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b24; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py tests/test_graph_gpu.py tests/test_decoder_gpu.py -x -q -m gpu > $out/tests.log 2>&1; echo "decoder tests rc=$?"; tail -5 $out/tests.log
AB_ARGS="--in-flight 1" tools/ab_bench.sh r4b24_ab1 default default+RAC_FUSE_SAMPLING_MIXING=0 default default+RAC_FUSE_SAMPLING_MIXING=0
AB_ARGS="" tools/ab_bench.sh r4b24_ab4 default default+RAC_FUSE_SAMPLING_MIXING=0 default default+RAC_FUSE_SAMPLING_MIXING=0
