#!/bin/bash
# round 3: parity of the mm kernels after the non-temporal r / f streams; DWC / unprotected (lane-replica panel kernel) A/B cached vs non-temporal
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3q
mkdir -p $OUT
timeout 500 python -m pytest tests -m gpu -x -q -k "test_mm or dropin or rejected" -p no:cacheprovider > $OUT/mm_tests.txt 2>&1; tail -3 $OUT/mm_tests.txt
for rep in 1 2; do
  for lib in cur gpurun_ab/lib_pnt0.so; do
    if [ "$lib" = "cur" ]; then unset COAST_LIB_OVERRIDE; else export COAST_LIB_OVERRIDE=$ROOT/$lib; fi
    echo "== $lib"; PERF_REPS=7 timeout 120 python tools/perf_kernels.py --only mm 2>/dev/null | grep mm256
  done
done 2>&1 | tee $OUT/panel_nt.txt
