cd /tmp && export TMPDIR=/tmp
for v in default gn32 gn64 gn80 gn112; do
  if [ $v = default ]; then unset RACFORMER_HIP_LIB; else export RACFORMER_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_$v.so; fi
  (cd $GRAFT_REPO_ROOT && timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/r5ag/$v -o run --output-format csv -- python3 tools/exp_gen_narrow.py > gpurun_out/r5ag_$v.log 2>&1) || { echo $v FAILED; tail -3 $GRAFT_REPO_ROOT/gpurun_out/r5ag_$v.log; }
  echo $v $(grep generator_ws $GRAFT_REPO_ROOT/gpurun_out/r5ag/$v/run_kernel_stats.csv | cut -d, -f1-4,6)
done
