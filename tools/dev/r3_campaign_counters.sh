#!/bin/bash
# round 3: campaigns aimed at the loop counters, which COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC puts inside the sphere of replication
OUT=gpurun_out/${1:-r3c2}
mkdir -p $OUT
python - > $OUT/campaign_counters.txt 2>&1 <<'PY'
import importlib.util, json, os, sys
root = os.getcwd()
spec = importlib.util.spec_from_file_location("coast_campaign", os.path.join(root, "tools", "campaign.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
import coast_amd
eng = coast_amd.Engine(0)
runs = os.environ.get("RUNS", "1000")
print("tools/campaign.py --counters-in-sor -t %s (one process, one engine): every upset hits a loop counter of the call" % runs)
print("%-12s %-5s %6s %8s %7s %7s %9s %9s %14s %10s" % ("benchmark", "mode", "runs", "success", "errors", "faults", "timeouts", "coverage", "TMR_ERROR_CNT", "engine"))
for b in (["mm", "--side", "9"], ["sha256"], ["aes"], ["crc16"], ["chsha"], ["cache_test"]):
    for m in ("TMR", "DWC", "NONE"):
        try:
            a = mod.parse(["-b"] + b + ["-m", m, "-t", runs, "--counters-in-sor", "-n"])
            rec, s = mod.run_campaign(a, eng)
            print("%-12s %-5s %6d %8d %7d %7d %9d %8.2f%% %14d %10s" % (b[0], m, s["runs"], s["success"], s["errors"], s["faults"], s["timeouts"],
                                                                 s["coverage_pct"], s["TMR_ERROR_CNT"], s["engine"]))
        except BaseException as e:
            print(b[0], m, "FAILED", repr(e))
        sys.stdout.flush()
PY
cat $OUT/campaign_counters.txt
