#!/bin/bash
# tools/next_round_first_call.sh -- what the round-4 sessions could no longer run (the GPU budget was spent): one gpurun call, ~6 min.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/next_round_first_call.sh'
# 1. the whole GPU suite (580 collected at the end of round 4; the last full run, 577 passed, predates test_campaign_physical_real_all_registers_mm256 and
#    test_mm_physical_upsets_of_the_shared_staging_registers_are_silent, which passed by itself);
# 2. campaign --reg-model physical-real-all in its two-launch form (written blind, exercised against a stand-in engine only):
#    does a replica-private class still show an error once the staging flips run in a launch of their own?
#    (profiles/r04_campaign_physical_real_all.txt: 4 of 3103 in the one-launch form);
# 3. the default bench line.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_first
mkdir -p $OUT
cd $ROOT
timeout 420 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/suite.txt
for seed in 0 1; do
  timeout 60 python tools/campaign.py -b mm --side 256 -m TMR -t 5000 --reg-model physical-real-all --seed $seed -n 2>&1 | cut -c1-1800 > $OUT/physical_real_all_seed$seed.txt
done
timeout 60 python tools/campaign.py -b mm --side 256 -m TMR -t 5000 --reg-model physical-real -n 2>&1 | cut -c1-1800 > $OUT/physical_real.txt
timeout 240 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/suite.txt; grep -h "acc \|b_frag\|a_frag\|s_raw\|f_raw\|Coverage" $OUT/physical_real_all_seed*.txt
