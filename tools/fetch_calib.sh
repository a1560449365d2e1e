#!/bin/bash
# tools/fetch_calib.sh -- run on the GPU box: FETCH_SIZE / WRITE_SIZE of tools/fetch_calib's four 1 GiB access patterns (separate --pmc passes)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fetch_calib
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d "$OUT/f" -o c -- $ROOT/tools/fetch_calib > "$OUT/run_f.txt" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/w" -o c -- $ROOT/tools/fetch_calib > "$OUT/run_w.txt" 2>&1
python - "$OUT" <<'PY'
import glob, os, sqlite3, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for db in glob.glob(os.path.join(out, "*", "*.db")):
    cur = sqlite3.connect(db).cursor()
    for k, c, v in cur.execute("select kernel_name,counter_name,value from counters_collection"):
        acc[k.split("(")[0]][c].append(v)
print("kernel (moves 2^30 bytes once)      counter      mean (KiB)     x1024 / 2^30")
for k in sorted(acc):
    for c, vs in sorted(acc[k].items()):
        m = sum(vs) / len(vs)
        print("%-34s %-12s %12.1f   %.4f" % (k, c, m, m * 1024 / 2**30))
PY
find "$OUT" -name "*.db" -delete
