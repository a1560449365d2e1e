#!/bin/bash
# round-3 development run: AES after the address / fold changes, the crc16 LDS/VALU hybrid variants, the typed exit vote
OUT=gpurun_out/${1:-r3d}
mkdir -p $OUT
K="aes or loop_counters or operand_type or common_mode or exit_vote or crc16_stream"
(timeout 600 python -m pytest tests -m gpu -q -k "$K" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -4 $OUT/pytest.log
for hyb in 0 16 12 8 6; do
  if [ $hyb = 0 ]; then unset COAST_CRC_HYB; else export COAST_CRC_HYB=$hyb; fi
  for rep in 1 2; do
    timeout 120 python bench.py --workload crc16 --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('crc16 hyb $hyb', 'kernel_ms %.3f frac %.4f ok %s corr %d' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['outputs_match_unprotected'], d['corrected_faults']))"
  done
done 2>&1 | tee $OUT/crc_hyb.txt
unset COAST_CRC_HYB
timeout 200 python tools/perf_kernels.py --only aes > $OUT/perf_aes.txt 2>&1; cat $OUT/perf_aes.txt | grep aes
for b in 0 16777216; do timeout 120 python bench.py --workload aes --batch $b --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('aes batch $b', 'ms/step %.4f kernel_ms %.4f frac %.4f ok %s det %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['outputs_match_unprotected'], d['dwc_detected']))"; done | tee $OUT/aes_bench.txt
