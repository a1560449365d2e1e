"""CPU tests of the drop-in boundary: libcoast_hip.so loads and exports every symbol include/coast_hip.h declares, the
binding's struct layouts match the header, bad arguments are rejected, and the product refuses to run without a GPU
(no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "coast_hip.h")


def _declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(coast_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from coast_amd import _lib

    lib = _lib.load()
    names = _declared_functions()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), "libcoast_hip.so does not export %s" % n
    assert set(names) == set(_lib.SYMBOLS), "python binding and header disagree"
    assert lib.coast_abi_version() == 8  # 8: cloned staging loads by default (COAST_F_SINGLE_STAGING), mm_mfma_blk4_kernel; 7: COAST_F_CLONE_STAGING, COAST_SITE_MM_PREG; 3: coast_stats.kernel_ms/.hbm_bytes, coast_launch_info; 4: control-flow signatures; 5: COAST_REPLICA_ALL, COAST_ETIMEOUT; 6: COAST_F_LOCAL_STORE_SYNC


def test_header_compiles_as_c_and_layouts_match(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "coast_hip.h"\n#include <stdio.h>\nint main(void){printf("%zu %zu %zu %zu\\n",'
                   'sizeof(coast_fault),sizeof(coast_cfg),sizeof(coast_stats),sizeof(coast_launch_info));return 0;}\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    from coast_amd import _lib

    assert sizes == [_lib.FAULT_DTYPE.itemsize, C.sizeof(_lib.CoastCfg), C.sizeof(_lib.CoastStats),
                     C.sizeof(_lib.CoastLaunchInfo)] == [16, 12, 48, 40]
    from oracle import oracle as orc

    assert orc.FAULT_DTYPE == _lib.FAULT_DTYPE  # the oracle and the product consume the same fault records


def test_dropin_exports_reference_names():
    path = os.path.join(ROOT, "coast_amd", "lib", "libcoast_dropin.so")
    if not os.path.exists(path):
        pytest.skip("libcoast_dropin.so not built")
    syms = subprocess.check_output(["nm", "-D", path], text=True)
    for name in ("crc16", "aes_enc_dec", "sha256_hash", "coast_dropin_matrix_multiply", "TMR_ERROR_CNT", "__SYNC_COUNT",
                 "FAULT_DETECTED_DWC"):
        assert re.search(r"\b%s\b" % re.escape(name), syms), name


def test_no_cpu_fallback():
    """Without a GPU the engine must refuse loudly; with one this test is vacuous."""
    import torch

    import coast_amd

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        coast_amd.Engine(0)
    with pytest.raises(Exception):
        coast_amd.crc16(b"abc")
    from coast_amd import _lib

    h = C.c_void_p()
    assert _lib.load().coast_create(C.byref(h), 0) != 0 and not h.value


def test_product_never_imports_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "coast_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".c", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f
                assert "coast_oracle.h" not in txt and "liboracle" not in txt, f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle import", bench)]
    assert uses, "bench.py's cpu_baseline leg should use the oracle"
    for u in uses:  # every use sits inside a cpu_baseline* function
        fn = re.findall(r"^def (\w+)", bench[:u], re.M)[-1]
        assert fn.startswith("cpu_baseline"), fn


def test_make_faults_layout():
    import coast_amd

    f = coast_amd.make_faults([(0x1122334455, 2, coast_amd.SITE_MM_OPB, 77, 31, 5)])
    raw = f.tobytes()
    assert len(raw) == 16
    assert int.from_bytes(raw[0:8], "little") == 0x1122334455 and int.from_bytes(raw[8:12], "little") == 77
    assert list(raw[12:16]) == [2, coast_amd.SITE_MM_OPB, 31, 5]


def test_c_abi_soak_tool_builds_against_the_header(tmp_path):
    """tools/dev/aes_soak.cpp (the crash soak behind DESIGN.md 8.6: coast_aes128_batch from a C host, no Python) compiles and links
    against include/coast_hip.h and the in-tree library -- no GPU is touched."""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    lib = os.path.join(root, "coast_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libcoast_hip.so")):
        pytest.skip("library not built")
    p = subprocess.run([hipcc, "-O2", "-std=c++17", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tools", "dev", "aes_soak.cpp"),
                        "-o", str(tmp_path / "aes_soak"), "-L", lib, "-lcoast_hip"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
