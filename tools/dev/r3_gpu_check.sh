#!/bin/bash
# round-3 development run on ONE box: the new parity tests, the default bench line, the crc16 workgroup shapes, perf lines
OUT=gpurun_out/${1:-r3b}
mkdir -p $OUT
K="votes_on or common_mode or loop_counters or replica0_counter or campaign_physical or rccl_path or are_rejected or crc16_faults_vs_oracle or sha256_faults_vs_oracle or aes_faults_vs_oracle or aes_bank or quicksort_vs_oracle or mm_256_register_block"
(timeout 600 python -m pytest tests -m gpu -q -x -k "$K" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -4 $OUT/pytest.log
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo bench rc=$?
for shape in default 768x4 768x3 512x6 512x4; do
  if [ $shape = default ]; then unset COAST_CRC_SHAPE; else export COAST_CRC_SHAPE=$shape; fi
  for rep in 1 2; do
    timeout 120 python bench.py --workload crc16 --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('crc16 shape $shape', 'kernel_ms %.3f frac %.4f ok %s corr %d' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['outputs_match_unprotected'], d['corrected_faults']))"
  done
done 2>&1 | tee $OUT/crc_shapes.txt
unset COAST_CRC_SHAPE
timeout 300 python tools/perf_kernels.py --only quicksort,chaes,crazycf,indexed,aes > $OUT/perf.txt 2>&1; tail -30 $OUT/perf.txt
