#!/usr/bin/env python3
"""Time every protected kernel at the BASELINE.json config sizes (1 GPU) with HIP events on the launch stream.
Not the bench contract (that is bench.py) -- a development probe whose output goes under profiles/."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coast_amd  # noqa: E402


def timeit(fn, reps=int(os.environ.get("PERF_REPS", "5")), warm=int(os.environ.get("PERF_WARM", "2"))):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    eng = coast_amd.Engine(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    res = {}

    def want(k):
        return not a.only or k in a.only.split(",")

    if want("mm"):
        for rep in (3, 2, 1):
            batch, n = 2048, 256
            f = torch.randint(-2**31, 2**31, (batch, n, n), dtype=torch.int32, device="cuda", generator=g)
            s = torch.randint(-2**31, 2**31, (batch, n, n), dtype=torch.int32, device="cuda", generator=g)
            r = torch.empty_like(f)
            cfg = coast_amd.XmrConfig(rep)
            mn, av = timeit(lambda: eng.mm_batch(f, s, out=r, cfg=cfg))
            res["mm256_rep%d" % rep] = {"ms": mn, "elems_per_s": batch * n * n / mn * 1e3,
                                        "TMAC_alg": batch * n**3 / mn * 1e-9,
                                        # the replica count's own int8 work (ten limb products per MAC and replica) against 5.0 POP/s
                                        "frac_of_int8_peak": 2.0 * batch * n**3 * 10 * rep / (mn * 1e-3) / 5.0e15,
                                        "tile": os.environ.get("COAST_MM_TILE", "default")}
            del f, s, r
    if want("sha"):
        nm = 1 << 22
        msgs = torch.randint(0, 256, (nm, 64), dtype=torch.uint8, device="cuda", generator=g)
        out = torch.empty((nm, 32), dtype=torch.uint8, device="cuda")
        for rep in (3, 2, 1):
            cfg = coast_amd.XmrConfig(rep)
            mn, av = timeit(lambda: eng.sha256_batch(msgs, 64, out=out, cfg=cfg))
            res["sha256_4Mx64B_rep%d" % rep] = {"ms": mn, "msgs_per_s": nm / mn * 1e3, "GBs_in": nm * 64 / mn * 1e-6}
    if want("aes"):
        n = 1 << 20
        st = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda", generator=g)
        key = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda", generator=g)
        for rep in (2, 3, 1):
            for d in (0, 1):
                cfg = coast_amd.XmrConfig(rep)
                mn, av = timeit(lambda: eng.aes128_batch(st, key, d, cfg=cfg))
                res["aes128_1M_rep%d_dir%d" % (rep, d)] = {"ms": mn, "blocks_per_s": n / mn * 1e3,
                                                          "GBs": n * 64 / mn * 1e-6}
    if want("crc"):
        nbytes = 1 << 30
        data = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda", generator=g)
        for bl in (256, 255):
            nb = nbytes // bl
            out = torch.empty(nb, dtype=torch.int16, device="cuda")
            for rep in (3, 2, 1):
                cfg = coast_amd.XmrConfig(rep)
                mn, av = timeit(lambda: eng.crc16_batch(data[: nb * bl], bl, out=out, cfg=cfg), reps=3, warm=1)
                res["crc16_1GiB_bl%d_rep%d" % (bl, rep)] = {"ms": mn, "GBs": nb * bl / mn * 1e-6,
                                                            "frac_of_8TBs": nb * bl / mn * 1e-6 / 8000}
    if want("vote"):
        # default (memory-replicated) mode: three unprotected mm launches on three copies + the exit vote
        batch, n = 2048, 256
        f = [torch.randint(-2**31, 2**31, (batch, n, n), dtype=torch.int32, device="cuda", generator=g)]
        s = [torch.randint(-2**31, 2**31, (batch, n, n), dtype=torch.int32, device="cuda", generator=g)]
        f += [f[0].clone(), f[0].clone()]
        s += [s[0].clone(), s[0].clone()]
        r = [torch.empty_like(f[0]) for _ in range(3)]
        out = torch.empty_like(f[0])
        one = coast_amd.XmrConfig(1)
        mn, av = timeit(lambda: eng.sync_copies(r, out=out))
        nbytes = r[0].numel() * 4
        res["exit_vote_3x512MiB"] = {"ms": mn, "GBs_read+write": 4 * nbytes / mn * 1e-6, "frac_of_8TBs": 4 * nbytes / mn * 1e-6 / 8000}

        def default_mode():
            for c in range(3):
                eng.mm_batch(f[c], s[c], out=r[c], cfg=one)
            eng.sync_copies(r, out=out)

        mn, av = timeit(default_mode)
        res["mm256_default_mode_memx3"] = {"ms": mn, "elems_per_s": batch * n * n / mn * 1e3}
    if want("chsha"):
        nm, ln = 1 << 17, 16384  # 2 GiB of messages of the benchmark's size (2 x 8192 bytes)
        msgs = torch.randint(0, 256, (nm, ln), dtype=torch.uint8, device="cuda", generator=g)
        for rep in (3, 2, 1):
            cfg = coast_amd.XmrConfig(rep)
            mn, av = timeit(lambda: eng.chsha_batch(msgs, ln, cfg=cfg))
            res["chsha_128Kx16KiB_rep%d" % rep] = {"ms": mn, "GBs": nm * ln / mn * 1e-6, "msgs_per_s": nm / mn * 1e3}
        del msgs
    if want("cache"):
        na, n = 1 << 20, 600
        arr = torch.arange(n, dtype=torch.int32, device="cuda").repeat(na, 1).contiguous()
        for rep in (3, 2, 1):
            cfg = coast_amd.XmrConfig(rep)
            mn, av = timeit(lambda: eng.cache_test_batch(arr, cfg=cfg))
            res["cache_test_1Mx600_rep%d" % rep] = {"ms": mn, "GBs": na * n * 4 / mn * 1e-6,
                                                    "frac_of_8TBs": na * n * 4 / mn * 1e-6 / 8000}
        del arr
    if want("quicksort"):
        na, n = 1 << 14, 580  # array_elements, tests/quicksort/quicksort.c:82
        src = torch.randint(-2**31, 2**31, (na, n), dtype=torch.int32, device="cuda", generator=g)
        work = torch.empty_like(src)
        for rep in (3, 2, 1):
            cfg = coast_amd.XmrConfig(rep)

            def qs():
                work.copy_(src)  # quick_sort works in place: the copy (38 MB at HBM rate) rides along
                eng.quicksort_batch(work, cfg=cfg)

            eng.reset_stats()
            mn, av = timeit(qs)
            res["quicksort_16Kx580_rep%d" % rep] = {"ms": mn, "arrays_per_s": na / mn * 1e3, "elems_per_s": na * n / mn * 1e3}
        del src, work
    if want("chaes"):
        n = 1 << 20
        for type_ in (128128, 256256):
            nk, nb = type_ // 1000 // 32, type_ % 1000 // 32
            st = torch.randint(0, 256, (n, 4 * nb), dtype=torch.uint8, device="cuda", generator=g)
            key = torch.randint(0, 256, (n, 4 * nk), dtype=torch.uint8, device="cuda", generator=g)
            for rep in (3, 2, 1):
                cfg = coast_amd.XmrConfig(rep)
                mn, av = timeit(lambda: eng.chaes_batch(st, key, type_, 0, cfg))
                res["chaes_%d_1M_rep%d" % (type_, rep)] = {"ms": mn, "blocks_per_s": n / mn * 1e3}
            del st, key
    if want("crazycf"):
        n = 1 << 18
        prm = torch.tensor([[42, 20, 10]], dtype=torch.int32, device="cuda").repeat(n, 1)  # crazyCF.c:36, 11, 41
        for cfcss in (True, False):
            mn, av = timeit(lambda: eng.crazycf_batch(prm, cfcss=cfcss))
            res["crazycf_256K_runs_%s" % ("cfcss" if cfcss else "bare")] = {"ms": mn, "runs_per_s": n / mn * 1e3}
    if want("indexed"):
        # the counters-in-the-SoR forms (COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC): the sync-point-parity kernels, one lane per
        # sequential walk -- what VERDICT r2 weak 11 asked to be measured
        flags = coast_amd.F_BRANCH_SYNC | coast_amd.F_ADDR_SYNC
        nm = 1 << 16
        msgs = torch.randint(0, 256, (nm, 64), dtype=torch.uint8, device="cuda", generator=g)
        mn, av = timeit(lambda: eng.sha256_batch(msgs, 64, cfg=coast_amd.XmrConfig(3, 0, flags)), reps=3, warm=1)
        res["sha256_64Kx64B_counters_in_sor"] = {"ms": mn, "msgs_per_s": nm / mn * 1e3}
        data = torch.randint(0, 256, (1 << 24,), dtype=torch.uint8, device="cuda", generator=g)
        nb = (1 << 24) // 255
        mn, av = timeit(lambda: eng.crc16_batch(data[: nb * 255], 255, cfg=coast_amd.XmrConfig(3, 0, coast_amd.F_BRANCH_SYNC)), reps=3, warm=1)
        res["crc16_16MiB_bl255_counters_in_sor"] = {"ms": mn, "GBs": nb * 255 / mn * 1e-6}
        batch, n = 4096, 32
        f = torch.randint(-2**31, 2**31, (batch, n, n), dtype=torch.int32, device="cuda", generator=g)
        s = torch.randint(-2**31, 2**31, (batch, n, n), dtype=torch.int32, device="cuda", generator=g)
        mn, av = timeit(lambda: eng.mm_batch(f, s, cfg=coast_amd.XmrConfig(3, 0, flags)), reps=3, warm=1)
        res["mm32_4096_counters_in_sor"] = {"ms": mn, "elems_per_s": batch * n * n / mn * 1e3, "votes_per_call": 34881 + 4 * n**3 + 3 * n * n}
        mn, av = timeit(lambda: eng.mm_batch(f, s, cfg=coast_amd.XmrConfig(3)), reps=3, warm=1)
        res["mm32_4096_default_schedule"] = {"ms": mn, "elems_per_s": batch * n * n / mn * 1e3}
        nb = 1 << 14
        st = torch.randint(0, 256, (nb, 16), dtype=torch.uint8, device="cuda", generator=g)
        ky = torch.randint(0, 256, (nb, 16), dtype=torch.uint8, device="cuda", generator=g)
        for d in (0, 1):
            mn, av = timeit(lambda: eng.aes128_batch(st, ky, d, cfg=coast_amd.XmrConfig(3, 0, flags)), reps=3, warm=1)
            res["aes128_16K_counters_in_sor_dir%d" % d] = {"ms": mn, "blocks_per_s": nb / mn * 1e3, "votes_per_block": (469 + 1818, 593 + 2516)[d] + 8}
        na, ne = 1 << 12, 600
        arr = torch.arange(ne, dtype=torch.int32, device="cuda").repeat(na, 1).contiguous()
        mn, av = timeit(lambda: eng.cache_test_batch(arr, cfg=coast_amd.XmrConfig(3, 0, flags)), reps=3, warm=1)
        res["cache_test_4Kx600_counters_in_sor"] = {"ms": mn, "GBs": na * ne * 4 / mn * 1e-6, "votes_per_array": 4 * ne + 5}
        nmc, lnc = 1 << 12, 1024
        cm = torch.randint(0, 256, (nmc, lnc), dtype=torch.uint8, device="cuda", generator=g)
        mn, av = timeit(lambda: eng.chsha_batch(cm, lnc, cfg=coast_amd.XmrConfig(3, 0, flags)), reps=3, warm=1)
        res["chsha_4Kx1KiB_counters_in_sor"] = {"ms": mn, "GBs": nmc * lnc / mn * 1e-6, "votes_per_transform": 166 + 432 + 5}
    for k, v in res.items():
        print(k, json.dumps(v))


if __name__ == "__main__":
    main()
