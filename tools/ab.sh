#!/bin/bash
# [BENCH_ARGS="--workload crc16 --block-len 255"] [REPS="1 2"] tools/ab.sh libA.so libB.so ... -- development: bench the mm headline with differently built libraries back to back on ONE box
# (boxes differ in sustained clock by several percent, so A/B numbers from different gpurun calls do not compare)
# (round 5: bench.py quotes the headline with COAST_F_CLONE_STAGING; BENCH_ARGS="--single-staging" benches the unflagged kernel)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in ${REPS:-1 2}; do
  for lib in "$@"; do
    if [ "$lib" = "cur" ]; then unset COAST_LIB_OVERRIDE; else export COAST_LIB_OVERRIDE=$ROOT/$lib; fi
    python $ROOT/bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-extra ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'ms/step %.3f kernel_ms %.3f frac %.4f ok %s corr %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['outputs_match_unprotected'], d['corrected_faults']))"
  done
done
