"""Build libcoast_hip.so (gfx950 code object + C ABI) and libcoast_dropin.so (reference-named C symbols) in-tree.

hipcc cross-compiles without a GPU, so this is also the driver's "does it build" check.
"""
from __future__ import annotations

import hashlib
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
UNITS = ("coast_hip", "mm_phys_instances")  # translation units of libcoast_hip.so


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


def source_hash() -> str:
    """Content hash of everything libcoast_hip.so / libcoast_dropin.so are compiled from.  It is compiled into the library
    (coast_source_hash()), so a stale binary is detected wherever the tree travels."""
    h = hashlib.sha256()
    for path in sources():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stamp_ok(want: str) -> bool:
    """the hash is a string constant inside the library: no side file to lose"""
    try:
        with open(LIB, "rb") as fh:
            return want.encode() in fh.read()
    except OSError:
        return False


def sources():
    out = [os.path.join(HERE, "..", "include", "coast_hip.h")]
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp", ".inc", ".c", ".h")):
            out.append(os.path.join(CSRC, f))
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    """(Re)build in-tree when the library is missing or was compiled from other sources.  One builder at a time: the N ranks of a
    multi-GPU launch all come through here, and on a stale tree they must not compile into the same file concurrently."""
    import fcntl

    try:
        os.makedirs(LIBDIR, exist_ok=True)
        lock = open(os.path.join(LIBDIR, ".build.lock"), "w")
    except OSError:
        # read-only install: nothing can be (re)built here; a current library is still usable, a stale one is refused below
        if os.path.exists(LIB) and _stamp_ok(source_hash()) and not force:
            return LIB
        raise
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_examples(force: bool, stale: bool, verbose: bool) -> None:
    """The C host examples (tests run them on the GPU box).  Best effort: libcoast_hip.so resolves RCCL at first use, so a host
    without RCCL development files or gcc must still be able to load the library -- a failed demo link is a warning."""
    import warnings

    inc = os.path.join(HERE, "..", "include")
    demo_src = os.path.join(HERE, "..", "examples", "host_c_demo.c")
    demo = os.path.join(HERE, "..", "examples", "host_c_demo")
    mg_src = os.path.join(HERE, "..", "examples", "multi_gpu_c_demo.c")
    mg = os.path.join(HERE, "..", "examples", "multi_gpu_c_demo")
    jobs = []
    if os.path.exists(demo_src) and os.path.exists(DROPIN_OBJ) and (force or _newer(demo, [demo_src, DROPIN_OBJ, LIB])):
        jobs.append(["gcc", "-std=gnu11", "-O2", "-Dside=4", demo_src, os.path.join(CSRC, "mm_glue.c"), DROPIN_OBJ,
                     "-L", LIBDIR, "-lcoast_hip", "-Wl,-rpath,$ORIGIN/../coast_amd/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", demo])
    if os.path.exists(mg_src) and (force or stale or _newer(mg, [mg_src, LIB])):
        jobs.append(["gcc", "-std=gnu11", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", inc, mg_src,
                     "-L", LIBDIR, "-lcoast_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl",
                     "-Wl,-rpath,$ORIGIN/../coast_amd/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", mg])
    for cmd in jobs:
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
        except (OSError, subprocess.CalledProcessError) as e:
            warnings.warn("coast_amd.build: example %s not built (%s); the library itself is unaffected" % (cmd[-1], e))


def _build_locked(force: bool, verbose: bool) -> str:
    want = source_hash()
    stale = force or not os.path.exists(LIB) or not _stamp_ok(want)
    if stale:
        # two translation units, compiled side by side (mm_phys_instances.inc: the physical-upset instantiations of the matrix-core
        # kernels are half of the compile time), linked into one library
        base = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC"]
        objs, procs = [], []
        for name in UNITS:
            obj = os.path.join(LIBDIR, name + ".o")
            cmd = base + ['-DCOAST_SOURCE_HASH="%s"' % want, "-c", os.path.join(CSRC, name + ".hip"), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            objs.append(obj)
            procs.append((cmd, subprocess.Popen(cmd)))
        for cmd, p in procs:
            if p.wait() != 0:
                for _, q in procs:
                    if q.poll() is None:
                        q.kill()
                raise subprocess.CalledProcessError(p.returncode, cmd)
        cmd = base + ['-DCOAST_SOURCE_HASH="%s"' % want, "-shared", "-o", LIB] + objs  # (the definition is unused by a link: it names the build in the process list)
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        for obj in objs:
            os.remove(obj)
    dropin_src = os.path.join(CSRC, "dropin.c")
    if os.path.exists(dropin_src) and (stale or not os.path.exists(DROPIN) or not os.path.exists(DROPIN_OBJ)):
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-I", os.path.join(HERE, "..", "include"), "-o", DROPIN,
               dropin_src, "-L", LIBDIR, "-lcoast_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=gnu11", "-I", os.path.join(HERE, "..", "include"), "-c",
                               dropin_src, "-o", DROPIN_OBJ])
    _build_examples(force, stale, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
