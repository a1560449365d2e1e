#!/bin/bash
# round-3 development run: memory-copies mode, typed vote, VALU mm / cache_test in-kernel hooks, quicksort status, crc16 residency probe
OUT=gpurun_out/${1:-r3e}
mkdir -p $OUT
K="memory_copies or store_data_sync_mode or operand_type or mm_faults_vs_oracle or cache_test or loop_counters or common_mode or quicksort or dropin or unmodified or flag_matrix or plain_c_host or campaign or smoke or mm_golden"
(timeout 900 python -m pytest tests -m gpu -q -k "$K" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -6 $OUT/pytest.log
# is the crc16 stream held up by HBM or by its lookup chains?  the same kernel on streams that fit the 256 MB Infinity Cache / the L2s
for batch in 33554432 1048576 262144 65536; do
  timeout 120 python bench.py --workload crc16 --batch $batch --steps 40 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('crc16 blocks $batch (%.0f MiB)' % ($batch*256/2**20), 'kernel_ms %.4f frac %.4f ok %s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['outputs_match_unprotected']))"
done 2>&1 | tee $OUT/crc_residency.txt
python -c "
import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
