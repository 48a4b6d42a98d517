#!/bin/bash
# value_proj with the L2 touch of the stage after next (default) against without (build/lib_vp_notouch.so); absmax without redundant atomics
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r4b22; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_fused_gpu.py -x -q -m gpu -k "value_proj or conv3x3 or temporal or composed" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
cd /tmp && export TMPDIR=/tmp
for v in default vp_notouch; do
  if [ $v = default ]; then unset RACFORMER_HIP_LIB; else export RACFORMER_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_$v.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_$v -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-stress --in-flight 1 --steps 10 > $out/bench_$v.json 2> $out/prof_$v.log || { echo "$v failed"; tail -3 $out/prof_$v.log; exit 1; }
  f=$(find $out/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "value_proj|absmax_kernel|conv_pack|regroup" $f | cut -d, -f1-4 | cut -c1-120
done
