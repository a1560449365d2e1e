#!/bin/bash
# development call: mm parity tests on the TSTORE + DUP kernel, uniform register-file campaign before / after, A/B of the builds
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_c2
mkdir -p $OUT
cd $ROOT
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mm" 2>&1 | tail -25 > $OUT/mm_tests.txt
tail -5 $OUT/mm_tests.txt
timeout 120 python tools/campaign.py -b mm --side 256 -m TMR -t 5000 --reg-model uniform -n 2>&1 | cut -c1-3000 > $OUT/uniform_dup.txt
COAST_LIB_OVERRIDE=$ROOT/gpurun_ab/lib_r4.so timeout 120 python tools/campaign.py -b mm --side 256 -m TMR -t 5000 --reg-model uniform -n 2>&1 | cut -c1-3000 > $OUT/uniform_r4.txt
grep -h "Errors\|Coverage\|Registers with" $OUT/uniform_dup.txt $OUT/uniform_r4.txt | cut -c1-400
REPS="1 2" bash tools/ab.sh gpurun_ab/lib_r4.so gpurun_ab/lib_ts.so cur gpurun_ab/lib_dupnoabuf.so 2>&1 | tee $OUT/ab.txt
