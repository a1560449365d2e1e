#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_c6
mkdir -p $OUT; cd $ROOT
COAST_LIB_OVERRIDE=$ROOT/gpurun_ab/lib_base.so timeout 500 python tools/dev/bisect_crash.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bisect.txt
