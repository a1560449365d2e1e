#!/bin/bash
# round 3: the one-permute lookup address of the bank-replicated AES kernels -- parity, kernel times, bench leg
mkdir -p gpurun_out/r3k
timeout 300 python -m pytest tests -m gpu -x -q -k "aes" > gpurun_out/r3k/aes_tests.txt 2>&1; tail -3 gpurun_out/r3k/aes_tests.txt
timeout 200 python tools/perf_kernels.py --only aes > gpurun_out/r3k/perf_aes.txt 2>&1; cat gpurun_out/r3k/perf_aes.txt
timeout 200 python bench.py --workload aes --steps 40 --warmup 5 > gpurun_out/r3k/bench_aes.json 2> gpurun_out/r3k/bench_aes.err; tail -c 1500 gpurun_out/r3k/bench_aes.json
timeout 200 python bench.py --workload aes --batch 16777216 --steps 10 --warmup 2 > gpurun_out/r3k/bench_aes16.json 2> gpurun_out/r3k/bench_aes16.err; tail -c 900 gpurun_out/r3k/bench_aes16.json
COAST_AES_TABLES=replicated timeout 200 python tools/perf_kernels.py --only aes > gpurun_out/r3k/perf_aes_tmr_rep.txt 2>&1; grep rep3 gpurun_out/r3k/perf_aes_tmr_rep.txt
