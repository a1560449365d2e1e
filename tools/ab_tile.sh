#!/bin/bash
# development: bench the mm headline with different COAST_MM_TILE values back to back on ONE box: tools/ab_tile.sh panel128 blocks3 blocks2 lanes
# (`blocks`, the one-wave-per-SIMD kernel of round 2, was retired in round 5 and is rejected by coast_mm_batch)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in ${REPS:-1 2}; do
  for t in "$@"; do
    COAST_MM_TILE=$t python $ROOT/bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', 'ms/step %.3f kernel_ms %.3f ok %s corr %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['outputs_match_unprotected'], d['corrected_faults']))"
  done
done
