#!/bin/bash
cd $GRAFT_REPO_ROOT
AB_ARGS="--in-flight 1" tools/ab_bench.sh r4b14_ab default build/lib_conv_wnt.so default build/lib_conv_wnt.so
