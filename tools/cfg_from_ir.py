#!/usr/bin/env python3
"""Derive the control-flow graph the CFCSS pass would see for a C file: `clang -O0 -emit-llvm` (the reference's flow compiles at
-O0 before opt, tests/makefiles/Makefile.common), basic blocks in module order, terminator successors in operand order, calls of
functions defined in the module -- then the two things the pass adds before it numbers the blocks (projects/CFCSS/CFCSS.cpp):
one "CFerrorHandler.<fn>" block at the end of every function (createErrorBlocks, :107-126) and the function FAULT_DETECTED_CFC
(insertErrorFunction, :88-105) at the end of the module.  Output: the dict coast_amd.cfcss.assign() takes, as JSON.

    python tools/cfg_from_ir.py /root/reference/tests/crazyCF/crazyCF.c > tests/golden/crazycf_cfg.json

Used to pin the hand-written graphs in coast_amd/csrc/crazycf_kernel.hip and oracle/cfcss_oracle.c (tests/test_cfcss_cpu.py)."""
import json
import re
import subprocess
import sys

CLANG = "/opt/rocm/lib/llvm/bin/clang"
SKIP, RET = 8, 16


def parse_ir(text):
    funcs = []  # (name, [block dict])
    cur = None
    for line in text.splitlines():
        m = re.match(r"^define .*@([\w.]+)\(", line)
        if m:
            cur = (m.group(1), [])
            funcs.append(cur)
            continue
        if cur is None:
            continue
        if line.startswith("}"):
            cur = None
            continue
        m = re.match(r"^([\w.]+):", line)
        if m:
            cur[1].append({"label": m.group(1), "succ": [], "calls": [], "ret": False})
            continue
        s = line.strip()
        if not s or s.startswith(";"):
            continue
        if not cur[1]:  # an unnamed entry block
            cur[1].append({"label": "entry", "succ": [], "calls": [], "ret": False})
        blk = cur[1][-1]
        m = re.search(r"\bcall\b.*@([\w.]+)\(", s)
        if m:
            blk["calls"].append(m.group(1))
        if s.startswith("br "):
            blk["succ"] = re.findall(r"label %([\w.]+)", s)
        elif s.startswith("switch "):
            blk["succ"] = re.findall(r"label %([\w.]+)", s)
            blk["_switch"] = True
        elif blk.get("_switch") and re.match(r"^i\d+ -?\d+, label %", s):
            blk["succ"] += re.findall(r"label %([\w.]+)", s)
        elif s.startswith("]"):
            blk.pop("_switch", None)
        elif s.startswith("ret "):
            blk["ret"] = True
    return funcs


def build_graph(funcs, main="main"):
    # what the pass adds
    funcs = [(n, b + [{"label": "CFerrorHandler." + n, "succ": [], "calls": [], "ret": False, "skip": True}]) for n, b in funcs]
    funcs.append(("FAULT_DETECTED_CFC", [{"label": "FAULT_DETECTED_CFC", "succ": [], "calls": [], "ret": False},
                                         {"label": "CFerrorHandler.FAULT_DETECTED_CFC", "succ": [], "calls": [], "ret": False,
                                          "skip": True}]))
    index, entry = {}, {}
    n = 0
    for fi, (name, blocks) in enumerate(funcs):
        entry[name] = n
        for b in blocks:
            index[(fi, b["label"])] = n
            n += 1
    g = {"n_nodes": n, "flags": [], "func": [], "succ": [], "calls": [], "main_func": [f[0] for f in funcs].index(main),
         "names": []}
    for fi, (name, blocks) in enumerate(funcs):
        for b in blocks:
            me = index[(fi, b["label"])]
            g["flags"].append((SKIP if b.get("skip") else 0) | (RET if b["ret"] else 0))
            g["func"].append(fi)
            g["succ"].append([index[(fi, s)] for s in b["succ"]])
            g["names"].append(name + ":" + b["label"])
            if not b.get("skip"):
                for c in b["calls"]:
                    if c in entry and c != "FAULT_DETECTED_CFC":
                        g["calls"].append([me, entry[c]])
    return g


def main():
    src = sys.argv[1]
    ir = subprocess.run([CLANG, "-O0", "-w", "-emit-llvm", "-S", "-fno-discard-value-names", src, "-o", "-"], check=True,
                        capture_output=True, text=True).stdout
    g = build_graph(parse_ir(ir))
    json.dump(g, sys.stdout, indent=1)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
