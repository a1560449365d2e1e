"""Control-flow signatures on the CPU: the oracle's restatement of the CFCSS pass (projects/CFCSS/CFCSS.cpp) against the host-only
compile-time half of the C ABI (coast_cfcss_assign -- no GPU involved), both against the graph clang's -O0 IR gives for
tests/crazyCF/crazyCF.c, and the oracle's crazyCF against outputs of the reference program itself (tests/golden)."""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rand_graph(rng):
    """a module-shaped graph: edges stay inside their function, never target its entry block; calls go to later functions"""
    nf = rng.randint(1, 3)
    sizes = [rng.randint(1, 14) for _ in range(nf)]
    entry, func, succ, flags, base = [], [], [], [], 0
    for f, sz in enumerate(sizes):
        entry.append(base)
        for _ in range(sz):
            func.append(f)
            cand = [base + j for j in range(1, sz)]
            k = rng.choice([0, 1, 1, 2, 2, 3]) if cand else 0
            succ.append([rng.choice(cand) for _ in range(k)])
            flags.append(16 if k == 0 else 0)
        if rng.random() < 0.3 and sz > 1:  # the function's error-handler block
            flags[base + sz - 1] = 8
            succ[base + sz - 1] = []
        base += sz
    calls = sorted((rng.randrange(0, entry[f]), entry[f]) for f in range(1, nf) for _ in range(rng.choice([0, 1, 1, 2])))
    return {"n_nodes": base, "flags": flags, "func": func, "succ": succ, "calls": calls, "main_func": 0}


def _check_invariant(t, calls=()):
    """docs/source/cfcss.rst: G ^ d (^ D at fan-in blocks) == s on every legal transition"""
    for p in range(t["n_nodes"]):
        if t["flags"][p] & 8:
            continue
        for s in t["succ"][p]:
            if t["flags"][s] & 8 or not t["flags"][s] & 2:
                continue
            x = t["sig"][p] ^ t["sig_diff"][s]
            if t["flags"][s] & 1:
                x ^= t["sig_adj"][p]
            assert x == t["sig"][s], (p, s)
    for c, (b, e) in enumerate(calls):
        if t["flags"][b] & 8 or t["flags"][e] & 8:
            continue
        x = t["sig"][b] ^ t["sig_diff"][e]
        if t["flags"][e] & 1:
            x ^= t["call_pre_adj"][c]
        assert x == t["sig"][e], ("call", b, e)


def test_glibc_rand_restatement_matches_libc(orc):
    libc = C.CDLL("libc.so.6")
    for seed in (1, 42, 0, 12345678, 0x7FFFFFFF, 0x80000000, 0xDEADBEEF, 0xFFFFFFFF):
        libc.srand(C.c_uint(seed))
        want = [libc.rand() for _ in range(400)]
        assert orc.glibc_rand_seq(seed, 400).tolist() == want, seed


def test_crazycf_graph_matches_clang_ir(orc):
    """the hand-written graphs (product: crazycf_kernel.hip; oracle: cfcss_oracle.c) against tools/cfg_from_ir.py's reading of
    `clang -O0 -emit-llvm crazyCF.c` (committed; regenerated and compared when the reference checkout is here)"""
    from coast_amd import cfcss

    want = json.load(open(os.path.join(GOLDEN, "crazycf_cfg.json")))
    for got in (cfcss.crazycf_graph(), orc.graph_to_dict(orc.crazycf_graph())):
        for k in ("n_nodes", "flags", "func", "succ", "main_func"):
            assert got[k] == want[k], k
        assert [list(c) for c in got["calls"]] == want["calls"]
    src = "/root/reference/tests/crazyCF/crazyCF.c"
    if os.path.exists(src) and os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
        import subprocess
        import sys

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "cfg_from_ir.py"), src], capture_output=True, text=True,
                             check=True).stdout
        assert json.loads(out) == want


def test_assign_product_equals_oracle_crazycf(orc):
    from coast_amd import cfcss

    t = cfcss.crazycf_tables()
    assert t == orc.tables_to_dict(orc.cfcss_assign(orc.crazycf_graph()))
    g = cfcss.crazycf_graph()
    _check_invariant(t, g["calls"])
    assert t["n_nodes"] == 28 + t["n_buffers"] and len(set(t["sig"])) == t["n_nodes"]
    assert t["sig"][:28] == sorted(t["sig"][:28]) and 0 not in t["sig"][:28]  # ascending over the blocks (std::set order)
    # the unseeded glibc sequence: the first signature drawn is rand() % 65536 of srand(1)
    first = int(orc.glibc_rand_seq(1, 1)[0]) % 65536
    assert first in t["sig"]


def test_assign_product_equals_oracle_random_graphs(orc):
    """buffer-block insertion (crazyCF needs none): hundreds of module-shaped graphs, identical tables from the two
    implementations -- different data structures, different rand() (libc's vs the restated one) -- and the run-time invariant"""
    from coast_amd import cfcss

    rng = random.Random(2024)
    nbuf = 0
    for _ in range(400):
        g = _rand_graph(rng)
        a = cfcss.assign(g)
        assert a == orc.tables_to_dict(orc.cfcss_assign(g))
        _check_invariant(a, g["calls"])
        nbuf += a["n_buffers"]
    assert nbuf > 500


def test_assign_rejects_bad_graphs():
    from coast_amd import _lib, cfcss

    with pytest.raises(_lib.CoastLibraryError):
        cfcss.assign({"n_nodes": 2, "flags": [0, 0], "func": [0, 0], "succ": [[5], []], "calls": [], "main_func": 0})
    with pytest.raises(_lib.CoastLibraryError):
        cfcss.assign({"n_nodes": 300, "flags": [0] * 300, "func": [0] * 300, "succ": [[]] * 300, "calls": [], "main_func": 0})
    assert _lib.load().coast_cfcss_assign(None, None) != 0


def test_oracle_crazycf_matches_reference_outputs(orc, golden):
    """the program's arithmetic: `Total` and the `total so far` line of crazyCF.c compiled unmodified (gen_golden.py), for the
    source's own constants and a grid of srand arguments / sizes that reaches every case of the switch"""
    assert golden["crazycf_stdout"] == "total so far: 27\nTotal = 7\n"
    grid = golden["crazycf_grid"]
    assert grid[0] == [42, 20, 7, 27, 1] and max(r[1] for r in grid) > 37
    prm = np.array([[np.int64(s).astype(np.int32) if s < 2**31 else np.int64(s - 2**32).astype(np.int32), n, 10]
                    for s, n, *_ in grid], dtype=np.int32)
    for cfcss in (True, False):
        res, st = orc.crazycf_batch(prm, cfcss=cfcss)
        assert not st.any()
        assert res["total"].tolist() == [r[2] for r in grid]
        assert res["printed"].tolist() == [r[3] for r in grid]
        assert res["n_prints"].tolist() == [r[4] for r in grid]
    for (s, n, tot, pr, npr), p in zip(grid, prm):
        assert orc.crazycf_plain(int(p[0]), n, 10) == (tot, pr, npr)
    if os.path.exists(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libcoast_ref.so")):
        for s, n, tot, pr, npr in grid[:20]:
            assert orc.ref_crazycf(s, n) == (tot, pr, npr)


def test_oracle_block_walk_equals_source_form(orc):
    """timesThroughWhile is a literal in the reference source (the shim cannot vary it): the block walk is pinned for other values
    by the oracle's statement of the source as written"""
    rng = np.random.default_rng(5)
    prm = np.stack([rng.integers(-2**31, 2**31, 300), rng.integers(0, 120, 300), rng.integers(-3, 60, 300)], axis=1).astype(np.int32)
    res, st = orc.crazycf_batch(prm)
    assert not st.any()
    for p, r in zip(prm, res):
        assert orc.crazycf_plain(int(p[0]), int(p[1]), int(p[2])) == (int(r["total"]), int(r["printed"]), int(r["n_prints"]))
    # transitions: 2 calls + 2 returns + fillArray (2 + 3 size) + main
    assert int(res["blocks"][0]) > 3 * int(prm[0][1])


def test_oracle_detects_illegal_jumps(orc):
    """what the signatures are for: a branch target flipped to the start of a block that is no legal successor is caught at
    that block's check (or lands in an error handler); flipped to another legal successor it passes (the paper's known limit)"""
    from coast_amd import make_faults

    t = orc.cfcss_assign(orc.crazycf_graph())
    td = orc.tables_to_dict(t)
    base, _ = orc.crazycf_batch([[42, 20, 10]])
    nblk = int(base["blocks"][0])
    rng = random.Random(9)
    rows, n = [], 600
    for q in range(n):
        rows.append((q, 0, orc.SITE_CFC_PC, rng.randrange(nblk), rng.randrange(0, 6)))
    prm = np.tile(np.array([[42, 20, 10]], np.int32), (n, 1))
    res, st = orc.crazycf_batch(prm, cfcss=True, faults=make_faults(rows), tables=t)
    res0, st0 = orc.crazycf_batch(prm, cfcss=False, faults=make_faults(rows), tables=t)
    assert (st == orc.CFC_DETECTED).sum() > n // 2
    # protected: a run that ends OK made only legal transitions afterwards -- and there are few of them
    ok = st == orc.CFC_OK
    assert ok.sum() < (st0 == orc.CFC_OK).sum()
    wrong0 = ((st0 == orc.CFC_OK) & (res0["total"] != 7)).sum()
    wrong = (ok & (res["total"] != 7)).sum()
    assert wrong0 > 20 and wrong < wrong0 // 4, (wrong0, wrong)
    # tracker upsets: any live bit of RTS is caught at the next checked block
    rows = [(q, 0, orc.SITE_CFC_RTS, 7 + q % 50, q % 16) for q in range(200)]
    _, st = orc.crazycf_batch(prm[:200], cfcss=True, faults=make_faults(rows), tables=t)
    assert (st == orc.CFC_DETECTED).all()
    rows = [(q, 0, orc.SITE_CFC_RTS, 7 + q % 50, 16 + q % 16) for q in range(200)]  # bits 16..31 are not part of the i16 global
    res, st = orc.crazycf_batch(prm[:200], cfcss=True, faults=make_faults(rows), tables=t)
    assert not st.any() and (res["total"] == 7).all()
    assert td["n_nodes"] >= 28


def test_library_exports_cfcss_symbols():
    from coast_amd import _lib

    lib = _lib.load()
    for name in ("coast_cfcss_assign", "coast_crazycf_graph", "coast_crazycf_tables", "coast_crazycf_batch"):
        assert hasattr(lib, name)
    assert C.sizeof(_lib.CoastCfcTables) == 8 + 3 * 512 + 256 + 4 * 257 + 2048 + 128 + 128
