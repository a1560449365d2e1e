#!/bin/bash
# round-3 evidence run on ONE box: targeted parity tests, the physical register campaign, rocprofv3 stats + PMC passes per workload
OUT=gpurun_out/${1:-r3c}
mkdir -p $OUT
K="common_mode or campaign_physical or rccl_path or aes_lean or loop_counters or replica0_counter"
(timeout 600 python -m pytest tests -m gpu -q -k "$K" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -4 $OUT/pytest.log
(for seed in 0 1; do timeout 300 python tools/campaign.py -b mm --side 256 -m TMR -t 5000 --reg-model physical --seed $seed -n; done
 timeout 300 python tools/campaign.py -b mm --side 256 -m TMR -t 5000 --reg-model sites -n) > $OUT/campaign_physical.txt 2>&1
tail -25 $OUT/campaign_physical.txt
bash tools/profile.sh r03_mm > /dev/null 2>&1
bash tools/profile.sh r03_crc16_256 --workload crc16 --block-len 256 > /dev/null 2>&1
bash tools/profile.sh r03_crc16_255 --workload crc16 --block-len 255 > /dev/null 2>&1
bash tools/profile.sh r03_sha256 --workload sha256 > /dev/null 2>&1
bash tools/profile.sh r03_aes --workload aes > /dev/null 2>&1
bash tools/profile.sh r03_aes16Mi --workload aes --batch 16777216 > /dev/null 2>&1
for t in mm crc16_256 crc16_255 sha256 aes aes16Mi; do echo "== $t"; head -12 gpurun_out/prof_r03_$t/summary.txt; done
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo bench rc=$?
