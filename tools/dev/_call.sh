#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_aes_every
mkdir -p $OUT; cd $ROOT
for ev in 5 9 15 25 1; do
 for rep in 1 2; do
  python bench.py --workload aes --steps 200 --warmup 20 --profile-every $ev 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('every $ev: ms_per_step %.4f kernel_ms %.4f ratio %.3f frac %.3f' % (d['ms_per_step'], r.get('kernel_ms', 0), d['ms_per_step']/max(r.get('kernel_ms',1e-9),1e-9), r['frac']))" | tee -a $OUT/aes_every.txt
 done
done
