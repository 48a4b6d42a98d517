#!/bin/bash
# usage (GPU box): tools/ab_value_proj.sh <tag> lib1.so lib2.so ...   ("default" = the in-tree build)
# tools/exp_value_proj.py under rocprofv3 --kernel-trace once per library build; prints the launch-duration quartiles of each
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
dbs=""
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = default ]; then unset RACFORMER_HIP_LIB; else export RACFORMER_HIP_LIB=$GRAFT_REPO_ROOT/$lib; fi
  (cd $GRAFT_REPO_ROOT && timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/$tag/$name -o run -- python3 tools/exp_value_proj.py run 40 > gpurun_out/$tag.$name.log 2>&1) || { echo "$name FAILED"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/$tag.$name.log; exit 1; }
  dbs="$dbs gpurun_out/$tag/$name/run_results.db"
done
cd $GRAFT_REPO_ROOT && python3 tools/exp_value_proj.py report $dbs
