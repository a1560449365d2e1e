#!/bin/bash
# round-3 full validation: the whole GPU suite (what the driver runs at round end), smoke, the default bench line
OUT=gpurun_out/${1:-r3n}
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -4 $OUT/pytest.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log)
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo bench rc=$?
python - <<'PY'
import json,sys
j=json.loads(open("gpurun_out/%s/bench.json" % (sys.argv[1] if len(sys.argv)>1 else "r3n")).read().strip().split("\n")[-1])
print("mm", j["value"], j["ms_per_step"], j["roofline"]["frac"], j.get("outputs_match_unprotected"))
for k,v in j.get("extra",{}).items():
    if isinstance(v,dict) and "roofline" in v:
        print(k, v.get("value"), v.get("ms_per_step"), v["roofline"].get("frac"), v.get("outputs_match_unprotected"), v.get("stepwise_blocks_last_launch"))
PY
