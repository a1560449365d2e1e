#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_c10
mkdir -p $OUT; cd $ROOT
REPS="1 2" bash tools/ab.sh gpurun_ab/lib_oldbase.so cur 2>&1 | tee $OUT/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $OUT/suite.txt
for cfg in "TMR:" "TMR:--clone-staging" "DWC:" "DWC:--clone-staging" "NONE:"; do
  mode=${cfg%%:*}; extra=${cfg#*:}
  timeout 400 python tools/campaign.py -b mm --side 256 -m $mode -t 5000 --reg-model uniform --sgpr run $extra -n 2>&1 | grep -v amdgpu.ids | cut -c1-8000 > $OUT/uniform_${mode}${extra}.txt
  echo "== $mode $extra"; grep -h "Successes\|Errors\|Faults\|Invalid\|Coverage\|Scalar\|Aborts" $OUT/uniform_${mode}${extra}.txt | cut -c1-300
done
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 2500 $OUT/bench_default.json
