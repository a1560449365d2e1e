#!/bin/bash
# round-3 closing run: the tests the earlier runs did not reach, the AES profiles after the round's kernel changes, the default bench line
OUT=gpurun_out/${1:-r3f}
mkdir -p $OUT
K="quicksort_vs_oracle or quicksort_reference_vectors or store_data_sync_mode or dropin_counters or memory_copies or operand_type or test_mm_faults_vs_oracle or cache_test_faults"
(timeout 420 python -m pytest tests -m gpu -q -x -k "$K" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -5 $OUT/pytest.log
bash tools/profile.sh r03_aes --workload aes > /dev/null 2>&1
bash tools/profile.sh r03_aes16Mi --workload aes --batch 16777216 > /dev/null 2>&1
for t in aes aes16Mi; do grep -E "aes128_(enc|dec)_rep" gpurun_out/prof_r03_$t/summary.txt | head -2; done
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo bench rc=$?
