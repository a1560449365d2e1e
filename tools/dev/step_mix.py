#!/usr/bin/env python3
"""step_mix.py <file.s> [symbol-substring] -- development: instruction mix of every 60-MFMA step body of the matrix-core kernels in a
-save-temps assembly file (VALU / SALU / LDS / VMEM / waits / lane moves per body), to see what a step issues besides its MFMAs."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else "mm_mfma_blk"
parts = re.split(r"\n(_ZN5coast\w+):[^\n]*\n", txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    if want not in name:
        continue
    lines = [ln.strip() for ln in body.split("\n")]
    lines = [ln for ln in lines if ln and not ln.startswith((";", ".", "//")) and not ln.split(";")[0].strip().endswith(":")]
    idx = [k for k, ln in enumerate(lines) if ln.startswith("v_mfma")]
    print(name, "mfma", len(idx), "instructions", len(lines))
    nb = len(idx) // 60
    for b in range(nb):
        seg = lines[idx[60 * b]:idx[60 * b + 59] + 1]
        c = collections.Counter()
        for ln in seg:
            op = ln.split()[0]
            key = ("mfma" if op.startswith("v_mfma") else "lane" if op.startswith(("v_readlane", "v_writelane")) else "valu" if op.startswith("v_")
                   else "wait" if op.startswith("s_waitcnt") else "nop" if op.startswith("s_nop") else "bar" if op.startswith("s_barrier")
                   else "br" if op.startswith(("s_cbranch", "s_branch")) else "salu" if op.startswith("s_") else "dsr" if op.startswith("ds_read")
                   else "dsw" if op.startswith("ds_write") else "vld" if op.startswith("buffer_load") else "vst" if op.startswith("buffer_store")
                   else "mem" if op.startswith(("scratch", "global", "flat")) else op)
            c[key] += 1
        print("  body %2d " % b + " ".join("%s %d" % kv for kv in sorted(c.items())))
