#!/bin/bash
# usage (GPU box, through gpurun): tools/gpu_final_profiles.sh <tag> pmc|bench
# everything the round's profiles/ directory holds, in two calls:
#   pmc   : kernel-trace stats of the default command (three plans in flight) AND of the same command with one plan in flight
#           (the run the bench's HIP-event kernel timings agree with), + PMC traffic passes (FETCH_SIZE and WRITE_SIZE separately, as
#           MI355X_MICROARCH.md prescribes; eager, one plan: counters serialise the kernels anyway) for both configs; summarise them
#           with tools/pmc_traffic.py afterwards
#   bench : the default bench (6-cam + 3-cam) -- run after the PMC summaries are in profiles/, so that roofline.traffic is
#           taken from counters of the very kernel version that is being timed
tag=$1; what=$2
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
if [ "$what" = bench ]; then
  timeout -k 10 500 python3 bench.py > $out/bench_f8.json 2> $out/bench_f8.err || exit 1
  timeout -k 10 500 python3 bench.py --config f8_3cam > $out/bench_f8_3cam.json 2> $out/bench_f8_3cam.err || exit 1
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof -o run --output-format csv -- python3 $root/bench.py --no-cpu-baseline --no-stress > $out/bench_prof.json 2> $out/prof.log || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof1 -o run --output-format csv -- python3 $root/bench.py --no-cpu-baseline --no-stress --in-flight 1 > $out/bench_prof1.json 2> $out/prof1.log || exit 1
for cfg in f8 f8_3cam; do
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_f_$cfg -o run --output-format csv -- python3 $root/bench.py --no-cpu-baseline --no-stress --no-graph --steps 4 --warmup 2 --config $cfg > /dev/null 2> $out/pmc_f_$cfg.log || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_w_$cfg -o run --output-format csv -- python3 $root/bench.py --no-cpu-baseline --no-stress --no-graph --steps 4 --warmup 2 --config $cfg > /dev/null 2> $out/pmc_w_$cfg.log || exit 1
done
cd $root
ls $out/prof $out/prof1 $out/pmc_f_f8 | head -20
