#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_c13
mkdir -p $OUT; cd $ROOT
REPS="1 2 3" bash tools/ab.sh gpurun_ab/lib_oldbase.so gpurun_ab/lib_w0.so gpurun_ab/lib_w2.so 2>&1 | tee $OUT/ab.txt
