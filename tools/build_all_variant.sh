#!/bin/bash
# usage: tools/build_all_variant.sh <name> "<extra hipcc flags>"  -- every source of libracformer_hip.so compiled with the extra flags
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/racformer_amd/csrc
out=$root/build/$name
mkdir -p $out
objs=""
pids=""
for f in $src/*.hip $src/capi.cpp; do
  b=$(basename $f); b=${b%.*}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -x hip -c $f -o $out/$b.o ) &
  objs="$objs $out/$b.o"
  if [ $(jobs -r | wc -l) -ge 8 ]; then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/lib_$name.so $objs
echo build/lib_$name.so
