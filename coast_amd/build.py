"""Build libcoast_hip.so (gfx950 code object + C ABI) and libcoast_dropin.so (reference-named C symbols) in-tree.

hipcc cross-compiles without a GPU, so this is also the driver's "does it build" check.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcoast_hip.so")
DROPIN = os.path.join(LIBDIR, "libcoast_dropin.so")
DROPIN_OBJ = os.path.join(LIBDIR, "coast_dropin.o")  # static object: link-time interposition needs a regular object
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libcoast_hip.so cannot be built (there is no CPU fallback)")


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def sources():
    out = [os.path.join(HERE, "..", "include", "coast_hip.h")]
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp", ".inc", ".c", ".h")):
            out.append(os.path.join(CSRC, f))
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sources()
    if force or _newer(LIB, srcs):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB,
               os.path.join(CSRC, "coast_hip.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    dropin_src = os.path.join(CSRC, "dropin.c")
    if os.path.exists(dropin_src) and (force or _newer(DROPIN, [dropin_src, LIB])):
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-I", os.path.join(HERE, "..", "include"), "-o", DROPIN,
               dropin_src, "-L", LIBDIR, "-lcoast_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=gnu11", "-I", os.path.join(HERE, "..", "include"), "-c",
                               dropin_src, "-o", DROPIN_OBJ])
    demo_src = os.path.join(HERE, "..", "examples", "host_c_demo.c")
    demo = os.path.join(HERE, "..", "examples", "host_c_demo")
    if os.path.exists(demo_src) and os.path.exists(DROPIN_OBJ) and (force or _newer(demo, [demo_src, DROPIN_OBJ, LIB])):
        subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-Dside=4", demo_src, os.path.join(CSRC, "mm_glue.c"), DROPIN_OBJ,
                               "-L", LIBDIR, "-lcoast_hip", "-Wl,-rpath,$ORIGIN/../coast_amd/lib",
                               "-Wl,-rpath,/opt/rocm/lib", "-o", demo])
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
