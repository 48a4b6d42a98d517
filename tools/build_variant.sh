#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>" file1.hip [file2.hip ...]
# NOPK=0 in the environment: WITHOUT the library's -packed-fp32-ops switch (the packed-FP32 diagnostic builds of DESIGN 3.12).
# An experimental build of libracformer_hip.so for A/B runs (RACFORMER_HIP_LIB, tools/ab_bench.sh): the named sources are
# compiled with the extra flags, every other object is taken from the in-tree build.  Output: build/lib_<name>.so
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/racformer_amd/csrc
out=$root/build/$name
mkdir -p $out
make -C $src -j8 > /dev/null
objs=""
for o in $src/*.o; do
  b=$(basename $o .o)
  skip=0
  for f in "$@"; do [ "$b" = "$(basename $f .hip)" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for f in "$@"; do
  b=$(basename $f .hip)
  nopk="-Xclang -target-feature -Xclang -packed-fp32-ops"; [ "${NOPK:-1}" = 0 ] && nopk=""
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DRAC_DIAGNOSTIC_BUILD $nopk $flags -c $src/$b.hip -o $out/$b.o
  objs="$objs $out/$b.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/lib_$name.so $objs
echo build/lib_$name.so
