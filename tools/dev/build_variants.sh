#!/bin/bash
# tools/dev/build_variants.sh name:"flags" ... -- development: build differently configured libcoast_hip.so files into gpurun_ab/ (for tools/ab.sh)
cd $(dirname $0)/../..
mkdir -p gpurun_ab
H=$(python -c "import coast_amd.build as b; print(b.source_hash())")
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCOAST_SOURCE_HASH="\"$H\"" $f -o gpurun_ab/lib_$n.so coast_amd/csrc/coast_hip.hip coast_amd/csrc/mm_phys_instances.hip > /tmp/build_$n.log 2>&1; echo "built $n rc=$?" ) &
done
wait
