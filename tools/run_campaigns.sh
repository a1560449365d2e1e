#!/bin/bash
# tools/run_campaigns.sh [runs] -- the campaign matrix behind profiles/r02_campaign_5000runs.txt (run on the GPU box via gpurun):
# registers x {NONE, TMR, DWC} for every benchmark (mm at side 256 on the matrix-core engine), memory section x {NONE, TMR lane
# engine, TMR default mode, DWC default mode} for the reference's MSP430 campaign workloads (MxM, CRC) + sha256 + aes.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
RUNS=${1:-5000}
ONLY=${2:-all}  # "new" = only the rows added in the second half of round 2 (mm 256 on the register-block kernel, chaes, crazycf)
OUT=$ROOT/gpurun_out/campaign
mkdir -p $OUT/logs
: > $OUT/summary.jsonl
run() { python $ROOT/tools/campaign.py -t $RUNS -l $OUT/logs "$@" 2>/dev/null | tail -1 >> $OUT/summary.jsonl; }
run -b crazycf -m NONE
run -b crazycf -m CFCSS
for m in NONE TMR DWC; do
  run -b chaes -m $m
  run -b chaes -m $m --chaes-type 256192
done
if [ "$ONLY" = "new" ]; then
  run -b mm -m TMR --side 256
else
for m in NONE TMR DWC; do
  run -b mm -m $m --side 256
  for b in mm sha256 aes crc16 chsha cache_test quicksort; do run -b $b -m $m; done
done
for b in mm crc16 sha256 aes; do
  run -b $b -m NONE -s memory
  run -b $b -m TMR -s memory --mem-mode nomemrep
  run -b $b -m TMR -s memory --mem-mode default
  run -b $b -m DWC -s memory --mem-mode default
done
fi
python - <<PY
import json
print("%-11s %-5s %-9s %-9s %6s %8s %7s %7s %9s %7s %9s  %s" % ("benchmark","mode","section","mem_mode","runs","success","faults","errors","timeouts","aborts","coverage","engine"))
for ln in open("$OUT/summary.jsonl"):
    s=json.loads(ln)
    print("%-11s %-5s %-9s %-9s %6d %8d %7d %7d %9d %7d %8.2f%%  %s%s" % (s["benchmark"]+("256" if s["benchmark"]=="mm" and s["engine"]=="matrix_core" else ""), s["mode"], s["section"], s["mem_mode"] or "-", s["runs"], s["success"], s["faults"], s["errors"], s["timeouts"], s["aborts"], s["coverage_pct"], s["engine"], "" if not s["stepwise_blocks"] else " (+%d stepwise tiles)" % s["stepwise_blocks"]))
PY
ls $OUT/logs | head -3; rm -f $OUT/logs/*.json  # the per-run records are large: keep the .log of each campaign in the merge-back
