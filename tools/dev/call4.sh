#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_c4
mkdir -p $OUT; cd $ROOT
REPS="1" bash tools/ab.sh gpurun_ab/lib_base.so gpurun_ab/lib_ring.so gpurun_ab/lib_d1ring.so gpurun_ab/lib_d1abuf.so gpurun_ab/lib_d1ringnc.so gpurun_ab/lib_d1rings.so gpurun_ab/lib_d1ringf.so gpurun_ab/lib_d2ring.so gpurun_ab/lib_base.so 2>&1 | tee $OUT/ab.txt
