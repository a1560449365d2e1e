#!/usr/bin/env python3
"""Instruction mix of the gfx950 kernels: hipcc -S on the unity source, then per-kernel opcode counts.

    python tools/instr_mix.py [kernel-name-substring ...]     (no argument: every kernel, summary line only)

Used to derive the instruction-mix ceilings bench.py prices the VALU-bound kernels against (sha256, aes) and to
look at the hand-scheduled mm step.  Development tool; nothing at run time depends on it."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = "/tmp/coast_instr_mix.s"


def disassemble():
    src = os.path.join(ROOT, "coast_amd", "csrc", "coast_hip.hip")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", ASM, src],
                          stderr=subprocess.DEVNULL)
    return ASM


def kernels(path):
    cur, body = None, {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            body[cur] = []
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
        if cur and re.match(r"^\t[a-z]", line) and not line.startswith("\t."):
            body[cur].append(line.split()[0])
    return body


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    pats = sys.argv[1:]
    body = kernels(disassemble())
    dm = demangle(list(body))
    for k, ops in body.items():
        name = dm.get(k, k)
        if not ops or (pats and not any(p in name for p in pats)):
            continue
        c = collections.Counter(ops)
        cls = collections.Counter()
        for op, n in c.items():
            key = ("mfma" if "mfma" in op else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
                   "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
            cls[key] += n
        print("%s\n   total %d  %s" % (name[:150], len(ops), dict(cls)))
        if pats:
            for op, n in c.most_common(40):
                print("      %-28s %d" % (op, n))


if __name__ == "__main__":
    main()
