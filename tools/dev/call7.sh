#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_c7
mkdir -p $OUT; cd $ROOT
export COAST_LIB_OVERRIDE=$ROOT/gpurun_ab/lib_base.so
timeout 200 python tools/dev/dbg2.py 2>&1 | grep -v amdgpu.ids | tee $OUT/dbg2.txt
timeout 300 python tools/campaign.py -b mm --side 256 -m TMR -t 5000 --reg-model uniform -n 2>&1 | grep -v amdgpu.ids | cut -c1-6000 > $OUT/uniform_base.txt
grep -h "Total runs\|Successes\|Errors\|Faults\|Invalid\|Coverage\|Registers with\|Scalar" $OUT/uniform_base.txt | cut -c1-900 | head -12
