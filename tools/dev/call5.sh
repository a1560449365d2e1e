#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_c5
mkdir -p $OUT; cd $ROOT
for lib in base d1abuf; do
  COAST_LIB_OVERRIDE=$ROOT/gpurun_ab/lib_$lib.so timeout 150 python tools/campaign.py -b mm --side 256 -m TMR -t ${RUNS:-5000} --reg-model uniform -n 2>&1 | grep -v amdgpu.ids | cut -c1-6000 > $OUT/uniform_$lib.txt
  echo "== $lib"; grep -h "Total runs\|Successes\|Errors\|Faults\|Coverage\|Registers with\|fault\|Error\|error" $OUT/uniform_$lib.txt | cut -c1-900 | head -12
done
