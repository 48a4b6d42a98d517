#!/bin/bash
# usage (GPU box): [AB_ARGS="--config f8_3cam"] tools/ab_bench.sh <tag> lib1.so lib2.so ...   ("default" = the in-tree build)
# one short bench run per library build (RACFORMER_HIP_LIB), prints the step time and the timed kernels of each
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  # "<lib>+VAR=value" runs that build with one extra environment variable (experiment switches)
  extra=""; case "$lib" in *+*) extra=${lib#*+}; lib=${lib%%+*};; esac
  name=$(basename $lib .so)${extra:+_$extra}
  if [ -n "$extra" ]; then export "$extra"; fi
  if [ "$lib" = default ]; then unset RACFORMER_HIP_LIB; else export RACFORMER_HIP_LIB=$GRAFT_REPO_ROOT/$lib; fi
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-stress --steps 30 $AB_ARGS > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -3 $out/$name.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("$out/$name.json")); r=d["roofline"]
print("$name", "samples/s", round(d["value"],2), "host_ms", round(d.get("host_enqueue_ms_per_step",0),2), "s4d_us", round(r["avg_launch_ms"]*1e3,1), "bev_us", round(r["bev_sampling"]["avg_launch_ms"]*1e3,1), " ".join(f"{k}={v['avg_launch_ms']*1e3:.1f}" for k,v in d["mfma"].items()))
PY
  if [ -n "$extra" ]; then unset "${extra%%=*}"; fi
done
