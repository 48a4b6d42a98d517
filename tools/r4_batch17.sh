#!/bin/bash
# four-wave generator: two workgroups per CU / one workgroup per CU (LDS padded), four plans in flight
cd $GRAFT_REPO_ROOT
AB_ARGS="" tools/ab_bench.sh r4b17_ab4 default build/lib_gen4.so build/lib_gen4pad.so default build/lib_gen4.so build/lib_gen4pad.so
AB_ARGS="--in-flight 1" tools/ab_bench.sh r4b17_ab1 build/lib_gen4pad.so
