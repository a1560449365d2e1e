#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_c3
mkdir -p $OUT; cd $ROOT
for lib in ts d1nt d2nt; do
  export COAST_LIB_OVERRIDE=$ROOT/gpurun_ab/lib_$lib.so
  echo "== $lib"; timeout 120 python tools/dev/dbg1.py 2>&1 | grep -v amdgpu.ids | tail -12
done 2>&1 | tee $OUT/dbg1.txt
REPS="1" bash tools/ab.sh gpurun_ab/lib_r4.so gpurun_ab/lib_d1nt.so gpurun_ab/lib_d2nt.so gpurun_ab/lib_d2ntna.so gpurun_ab/lib_d2ntnc.so gpurun_ab/lib_d2nts.so gpurun_ab/lib_d2ntf.so gpurun_ab/lib_r4.so 2>&1 | tee $OUT/ab.txt
