#!/bin/bash
# round 3: rocprofv3 evidence for the AES legs after the one-permute lookup address (1 Mi and 16 Mi blocks), TMR parity on the replicated tables
mkdir -p gpurun_out/r3l
timeout 300 python -m pytest tests -m gpu -x -q -k "aes" > gpurun_out/r3l/aes_tests.txt 2>&1; tail -3 gpurun_out/r3l/aes_tests.txt
bash tools/profile.sh r03b_aes --workload aes > /dev/null 2>&1
bash tools/profile.sh r03b_aes16Mi --workload aes --batch 16777216 > /dev/null 2>&1
for t in aes aes16Mi; do echo "== $t"; head -14 gpurun_out/prof_r03b_$t/summary.txt; grep -E "rep_kernel.*(SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_INSTS_LDS|SQ_ACTIVE_INST_LDS|SQ_LDS_BANK|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY)" gpurun_out/prof_r03b_$t/summary.txt; done
timeout 200 python tools/perf_kernels.py --only aes > gpurun_out/r3l/perf_aes.txt 2>&1; cat gpurun_out/r3l/perf_aes.txt
