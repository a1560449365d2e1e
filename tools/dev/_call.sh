#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_aes_fold3
mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "aes" 2>&1 | tail -4 | tee $OUT/tests.txt
for rep in 1 2; do
 for fold in 1 0; do
  COAST_AES_FOLD=$fold python bench.py --workload aes --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('fold $fold: ms_per_step %.4f kernel_ms %.4f ratio %.3f frac %.3f value %.3e' % (d['ms_per_step'], r.get('kernel_ms', 0), d['ms_per_step']/max(r.get('kernel_ms',1e-9),1e-9), r['frac'], d['value']))" | tee -a $OUT/aes_fold.txt
 done
done
