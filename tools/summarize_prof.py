#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory (rocprofv3 rocpd SQLite databases): per-kernel stats and the
per-dispatch mean of every PMC counter for the coast:: kernels."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:64]


for db in sorted(glob.glob(os.path.join(out, "trace*", "*.db"))):
    cur = sqlite3.connect(db).cursor()
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(db, out))
    print("   %-64s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("   %-64s %6d %12.1f %12.2f %7.2f" % (short(name), calls, total, avg, pct))

traffic = defaultdict(dict)
for db in sorted(glob.glob(os.path.join(out, "pmc_*", "*.db"))):
    cur = sqlite3.connect(db).cursor()
    acc = defaultdict(lambda: defaultdict(list))
    for k, c, v, d in cur.execute("select kernel_name,counter_name,value,duration from counters_collection"):
        if "coast::" in k:
            acc[short(k)][c].append((v, d))
    print("== PMC:", os.path.relpath(db, out))
    for k, d in acc.items():
        for cn, vs in d.items():
            mean = sum(v for v, _ in vs) / len(vs)
            print("   %-64s %-22s mean=%.6g n=%d mean_dur_us=%.1f" % (k, cn, mean, len(vs),
                                                                     sum(t for _, t in vs) / len(vs) / 1e3))
            if cn in ("FETCH_SIZE", "WRITE_SIZE"):
                traffic[k][cn] = mean

# HBM bytes per launch, corrected as MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE / WRITE_SIZE are in KiB and
# on gfx950 FETCH_SIZE reports exactly half of a wide coalesced streaming read -> double it; WRITE_SIZE as reported.
print("== HBM traffic per launch (FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 bytes)")
for k, d in traffic.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        print("   %-64s read=%.4g GB written=%.4g GB total=%.6g bytes" % (k, d["FETCH_SIZE"] * 2048e-9, d["WRITE_SIZE"] * 1024e-9,
                                                                         d["FETCH_SIZE"] * 2048 + d["WRITE_SIZE"] * 1024))
