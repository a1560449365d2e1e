#!/bin/bash
# tools/profile.sh <tag> [bench.py args...] -- run on the GPU box (via gpurun): rocprofv3 kernel stats + HBM PMC passes
# for bench.py.  Outputs under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra $*"
echo "$BENCH" > "$OUT/command.txt"
# 1) un-profiled reference line
$BENCH > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
# 2) kernel trace + stats
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $BENCH > "$OUT/bench_traced.json" 2> "$OUT/trace.err"
# 3) PMC passes, each its own run (TCC: FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- $BENCH > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    -d "$OUT/pmc_sq" -o bench -- $BENCH > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE \
    -d "$OUT/pmc_lds" -o bench -- $BENCH > /dev/null 2> "$OUT/pmc_lds.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d "$OUT/pmc_mfma" -o bench -- $BENCH > /dev/null 2> "$OUT/pmc_mfma.err"
python $ROOT/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt" | head -60
# the raw databases are large: keep only the summaries in the merge-back
find "$OUT" -name "*.db" -delete
