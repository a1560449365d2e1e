#!/usr/bin/env python3
"""fuzz_parity.py -- randomized GPU-vs-oracle soak: random shapes, modes, sync granularities and fault lists with heavy
collisions (several faults on one item / replica / step).  Every case must match the oracle bit for bit: outputs,
errors_corrected, sync_count, dwc_detected and the per-item flags.  Usage: tests/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coast_amd  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (test infrastructure: this tool is a test)


def dev(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        return torch.from_numpy(a.view(np.int32)).cuda()
    return torch.from_numpy(a).cuda()


def faults(rng, k, nitems, nrep, sites, max_step, max_index, hot):
    rows = []
    for _ in range(k):
        item = int(rng.choice(hot)) if hot is not None and rng.random() < 0.5 else int(rng.integers(0, nitems))
        rows.append((item, int(rng.integers(0, nrep)), int(rng.choice(sites)), int(rng.integers(0, max_step + 1)),
                     int(rng.integers(0, 32)), int(rng.integers(0, max_index))))
    return coast_amd.make_faults(rows)


def stats3(st):
    return {k: st[k] for k in ("errors_corrected", "sync_count", "dwc_detected")}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    eng = coast_amd.Engine(0)
    t0 = time.time()
    cases = {"mm": 0, "sha256": 0, "aes": 0, "crc16": 0, "cache_test": 0, "chsha": 0, "chaes_walk": 0, "crazycf_xmr": 0, "walks": 0}
    kinds = [k for k in os.environ.get("FUZZ_KINDS", "").split(",") if k in cases] or list(cases)  # e.g. FUZZ_KINDS=mm: one family only
    while time.time() - t0 < budget:
        kind = str(rng.choice(kinds))
        rep = int(rng.choice([1, 2, 3]))
        nrep = rep
        if kind == "mm":
            n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 19, 31, 32, 33, 48, 64, 65, 96, 128, 256]))
            batch = int(rng.integers(1, max(2, 20000 // (n * n)) + 1)) if n < 128 else 1
            sync_every = int(rng.choice([0, 0, 1, 2, 7, n]))
            clone = 0
            if n == 256:  # the matrix-core kernels: several matrices per workgroup group, with and without the cloned staging loads (round 5)
                batch, sync_every = int(rng.integers(1, 7)) if rng.random() < 0.8 else int(rng.integers(60, 70)), int(rng.choice([0, 0, 0, 16]))
                if rng.random() < 0.04:  # round 6: more matrices than mm_mfma_blk4_kernel has panel pairs -- workgroups with a second item (the f
                    batch = int(rng.integers(129, 136))  # panel replaced region by region across the hand-over), ~4 s of oracle each
                clone = 0 if sync_every == 0 and rng.random() < 0.5 else coast_amd.F_SINGLE_STAGING  # (cloned staging loads are the default)
            f = rng.integers(0, 2**32, (batch, n, n), dtype=np.uint32)
            s = rng.integers(0, 2**32, (batch, n, n), dtype=np.uint32)
            nit = batch * n * n
            hot = rng.integers(0, nit, 3)
            fl = faults(rng, int(rng.integers(0, 60)) if rep > 1 else 0, nit, nrep, [0, 1, 2], n, 1, hot)
            exp, est, edet = orc.mm_xmr(f, s, replicas=rep, sync_every=sync_every, faults=fl)
            det = torch.zeros(nit, dtype=torch.uint8, device="cuda")
            eng.reset_stats()
            eng.inject_faults(fl)
            got = eng.mm_batch(dev(f), dev(s), cfg=coast_amd.XmrConfig(rep, sync_every, clone), detected=det).cpu().numpy().view(np.uint32)
            ok = (got == exp).all() and stats3(eng.stats()) == est and (det.cpu().numpy() == edet).all()
            desc = "mm n=%d batch=%d rep=%d V=%d k=%d clone=%d" % (n, batch, rep, sync_every, len(fl), int(not clone))
        elif kind == "sha256":
            ln = int(rng.choice([0, 1, 3, 8, 55, 56, 57, 63, 64, 65, 100, 119, 120, 128, 200, 300]))
            stride = ln + int(rng.choice([0, 0, 1, 3, 4])) if rng.random() < 0.5 else ((ln + 15) // 16) * 16 + 16 * int(rng.integers(0, 2))
            stride = max(stride, 1)
            nm = int(rng.integers(1, 400))
            msgs = rng.integers(0, 256, (nm, stride), dtype=np.uint8)
            ncomp = ln // 64 + (1 if ln % 64 < 56 else 2)
            hot = rng.integers(0, nm, 3)
            rows = []
            for _ in range(int(rng.integers(0, 80)) if rep > 1 else 0):
                site = int(rng.choice([8, 9, 10]))
                step = int(rng.integers(0, ncomp + 1)) if site == 10 else int(rng.integers(0, ncomp * 64))
                item = int(rng.choice(hot)) if rng.random() < 0.5 else int(rng.integers(0, nm))
                rows.append((item, int(rng.integers(0, nrep)), site, step, int(rng.integers(0, 32)), int(rng.integers(0, 8))))
            fl = coast_amd.make_faults(rows)
            exp, est, edet = orc.sha256_xmr(msgs, ln, replicas=rep, faults=fl)
            det = torch.zeros(nm, dtype=torch.uint8, device="cuda")
            eng.reset_stats()
            eng.inject_faults(fl)
            got = eng.sha256_batch(torch.from_numpy(msgs).cuda(), ln, cfg=coast_amd.XmrConfig(rep), detected=det).cpu().numpy()
            ok = (got == exp).all() and stats3(eng.stats()) == est and (det.cpu().numpy() == edet).all()
            desc = "sha256 len=%d stride=%d nm=%d rep=%d k=%d" % (ln, stride, nm, rep, len(fl))
        elif kind == "aes":
            n = int(rng.integers(1, 600))
            d = int(rng.integers(0, 2))
            sync_every = int(rng.choice([0, 0, 1]))
            st = rng.integers(0, 256, (n, 16), dtype=np.uint8)
            key = rng.integers(0, 256, (n, 16), dtype=np.uint8)
            hot = rng.integers(0, n, 3)
            fl = faults(rng, int(rng.integers(0, 80)) if rep > 1 else 0, n, nrep, [16, 17], 10, 4, hot)
            es, ek, est, edet = orc.aes128_xmr(st, key, d, replicas=rep, sync_every=sync_every, faults=fl)
            ds, dk = torch.from_numpy(st.copy()).cuda(), torch.from_numpy(key.copy()).cuda()
            det = torch.zeros(n, dtype=torch.uint8, device="cuda")
            eng.reset_stats()
            eng.inject_faults(fl)
            eng.aes128_batch(ds, dk, d, cfg=coast_amd.XmrConfig(rep, sync_every), detected=det)
            ok = ((ds.cpu().numpy() == es).all() and (dk.cpu().numpy() == ek).all() and stats3(eng.stats()) == est
                  and (det.cpu().numpy() == edet).all())
            desc = "aes n=%d dir=%d rep=%d V=%d k=%d" % (n, d, rep, sync_every, len(fl))
        elif kind == "chsha":
            ln = 64 * int(rng.choice([0, 1, 2, 3, 5, 8, 16]))
            stride = ln + int(rng.choice([0, 0, 1, 3, 4, 16]))
            nm = int(rng.integers(1, 200))
            flags = int(rng.choice([0, 0, 1]))
            msgs = rng.integers(0, 256, (nm, max(stride, 1)), dtype=np.uint8)
            ncomp = ln // 64 + 1
            hot = rng.integers(0, nm, 3)
            rows = []
            for _ in range(int(rng.integers(0, 60)) if rep > 1 else 0):
                site = int(rng.choice([40, 41, 42]))
                step = int(rng.integers(0, ncomp)) if site == 42 else int(rng.integers(0, ncomp * 80))
                item = int(rng.choice(hot)) if rng.random() < 0.5 else int(rng.integers(0, nm))
                rows.append((item, int(rng.integers(0, nrep)), site, step, int(rng.integers(0, 32)), int(rng.integers(0, 5))))
            fl = coast_amd.make_faults(rows)
            exp, est, edet = orc.chsha_xmr(msgs, ln, replicas=rep, faults=fl, flags=flags)
            det = torch.zeros(nm, dtype=torch.uint8, device="cuda")
            eng.reset_stats()
            eng.inject_faults(fl)
            got = eng.chsha_batch(torch.from_numpy(msgs).cuda(), ln, cfg=coast_amd.XmrConfig(rep, 0, flags), detected=det)
            ok = ((got.cpu().numpy().view(np.uint32) == exp).all() and stats3(eng.stats()) == est
                  and (det.cpu().numpy() == edet).all())
            desc = "chsha len=%d stride=%d nm=%d rep=%d flags=%d k=%d" % (ln, stride, nm, rep, flags, len(fl))
        elif kind == "chaes_walk":  # CHStone aes statement by statement: any of the nine sizes, any combination of the counter flags
            type_ = int(rng.choice([128, 192, 256])) * 1000 + int(rng.choice([128, 192, 256]))
            nk, nb, nr = orc.chaes_geom(type_)
            n = int(rng.integers(1, 40))
            d = int(rng.integers(0, 2))
            flags = int(rng.choice([2, 4, 6, 6, 6 | 8, 6 | 16, 4 | 8 | 16, 6 | 64, 6 | 64, 6 | 64 | 1, 6 | 1]))
            st = rng.integers(0, 256, (n, 4 * nb), dtype=np.uint8)
            ky = rng.integers(0, 256, (n, 4 * nk), dtype=np.uint8)
            hot = rng.integers(0, n, 3)
            rows = []
            for _ in range(int(rng.integers(0, 60)) if rep > 1 else 0):
                item = int(rng.choice(hot)) if rng.random() < 0.5 else int(rng.integers(0, n))
                site = int(rng.choice([64, 65, 66, 67, 68, 66, 67, 68]))
                if site == 64:
                    step, bit, idx = int(rng.integers(0, nr + 2)), int(rng.integers(0, 32)), int(rng.integers(0, nb))
                elif site == 65:
                    step, bit, idx = int(rng.integers(0, nb * (nr + 1))), int(rng.integers(0, 32)), 0
                else:
                    step, idx = int(rng.integers(0, 700)), 0
                    bit = int(rng.integers(0, 4)) if rng.random() < 0.7 else int(rng.integers(0, 32))
                rows.append((item, int(rng.integers(0, nrep)), site, step, bit, idx))
            fl = coast_amd.make_faults(rows)
            exp, est, edet = orc.chaes_xmr(st, ky, type_, d, replicas=rep, flags=flags, faults=fl)
            ds, dk = torch.from_numpy(st.copy()).cuda(), torch.from_numpy(ky).cuda()
            det = torch.zeros(n, dtype=torch.uint8, device="cuda")
            eng.reset_stats()
            eng.inject_faults(fl)
            eng.chaes_batch(ds, dk, type_, d, coast_amd.XmrConfig(rep, 0, flags), detected=det)
            ok = (ds.cpu().numpy() == exp).all() and stats3(eng.stats()) == est and (det.cpu().numpy() == edet).all()
            desc = "chaes_walk type=%d n=%d dir=%d rep=%d flags=%d k=%d" % (type_, n, d, rep, flags, len(fl))
        elif kind == "crazycf_xmr":  # crazyCF under -TMR / -DWC
            n = int(rng.integers(1, 80))
            flags = int(rng.choice([0, 2, 4, 6, 6 | 16, 6 | 64, 6 | 64 | 1, 6 | 1]))
            prm = np.stack([rng.integers(0, 2**31 - 1, n), rng.integers(0, 60, n), rng.integers(0, 20, n)], axis=1).astype(np.int32)
            hot = rng.integers(0, n, 3)
            rows = []
            for _ in range(int(rng.integers(0, 60)) if rep > 1 else 0):
                item = int(rng.choice(hot)) if rng.random() < 0.5 else int(rng.integers(0, n))
                bit = int(rng.integers(0, 5)) if rng.random() < 0.7 else int(rng.integers(0, 32))
                rows.append((item, int(rng.integers(0, nrep)), int(rng.choice([72, 73, 74, 75])),
                             int(rng.integers(0, 2 * (int(prm[item, 1]) + int(prm[item, 2])) + 4)), bit, 0))
            fl = coast_amd.make_faults(rows)
            eres, estat, est, edet = orc.crazycf_xmr(prm, rep, flags, fl)
            det = torch.zeros(n, dtype=torch.uint8, device="cuda")
            eng.reset_stats()
            eng.inject_faults(fl)
            res, status = eng.crazycf_xmr_batch(torch.from_numpy(prm).cuda(), coast_amd.XmrConfig(rep, 0, flags), detected=det)
            g = res.cpu().numpy()
            ok = ((g[:, 0] == eres["total"]).all() and (g[:, 1] == eres["printed"]).all() and (g[:, 2] == eres["n_prints"].astype(np.int32)).all()
                  and (g[:, 3] == eres["blocks"].astype(np.int32)).all() and (status.cpu().numpy() == estat).all()
                  and stats3(eng.stats()) == est and (det.cpu().numpy() == edet).all())
            desc = "crazycf_xmr n=%d rep=%d flags=%d k=%d" % (n, rep, flags, len(fl))
        elif kind == "walks":  # the statement-by-statement forms (counter flags) of mm / aes / crc16 / cache_test / chsha / sha256
            which = str(rng.choice(["mm", "aes", "crc16", "cache_test", "chsha", "sha256"]))
            B, A, NL, NS, ND, L, O0 = 2, 4, 8, 16, 1, 64, 128
            flags = int(rng.choice([B, A, B | A, B | A, B | A | NL, B | A | NS, B | A | NL | NS, B | A | ND, B | A | L, B | A | L, B | A | L | ND]))
            nf = int(rng.integers(0, 40)) if rep > 1 else 0
            if which == "mm":
                n = int(rng.choice([1, 2, 3, 5, 8, 9, 12]))
                batch = int(rng.integers(1, 6))
                f = rng.integers(0, 2**32, (batch, n, n), dtype=np.uint32)
                sm = rng.integers(0, 2**32, (batch, n, n), dtype=np.uint32)
                nconds = (n + 1) * (n * n + n + 1)
                rows = [(int(rng.integers(0, batch)) * n * n, int(rng.integers(0, nrep)), int(rng.choice([0, 3, 4, 5])),
                         int(rng.integers(0, nconds)), int(rng.integers(0, 32)), 0) for _ in range(nf)]
                fl = coast_amd.make_faults(rows)
                exp, est, edet = orc.mm_xmr(f, sm, replicas=rep, flags=flags, faults=fl)
                det = torch.zeros(batch * n * n, dtype=torch.uint8, device="cuda")
                eng.reset_stats()
                eng.inject_faults(fl)
                got = eng.mm_batch(dev(f), dev(sm), cfg=coast_amd.XmrConfig(rep, 0, flags), detected=det).cpu().numpy().view(np.uint32)
                ok = (got == exp).all() and stats3(eng.stats()) == est and (det.cpu().numpy() == edet).all()
            elif which == "aes":
                n, d = int(rng.integers(1, 50)), int(rng.integers(0, 2))
                st = rng.integers(0, 256, (n, 16), dtype=np.uint8)
                key = rng.integers(0, 256, (n, 16), dtype=np.uint8)
                rows = []
                for _ in range(nf):
                    site = int(rng.choice([16, 17, 18, 19, 18, 19]))
                    rows.append((int(rng.integers(0, n)), int(rng.integers(0, nrep)), site,
                                 int(rng.integers(0, 11)) if site < 18 else int(rng.integers(0, 560)),
                                 int(rng.integers(0, 32)) if site < 18 else int(rng.integers(0, 8)), int(rng.integers(0, 4))))
                fl = coast_amd.make_faults(rows)
                es, ek, est, edet = orc.aes128_xmr(st, key, d, replicas=rep, faults=fl, flags=flags)
                ds, dk = torch.from_numpy(st.copy()).cuda(), torch.from_numpy(key.copy()).cuda()
                det = torch.zeros(n, dtype=torch.uint8, device="cuda")
                eng.reset_stats()
                eng.inject_faults(fl)
                eng.aes128_batch(ds, dk, d, cfg=coast_amd.XmrConfig(rep, 0, flags), detected=det)
                ok = ((ds.cpu().numpy() == es).all() and (dk.cpu().numpy() == ek).all() and stats3(eng.stats()) == est
                      and (det.cpu().numpy() == edet).all())
            elif which == "crc16":
                bl, nb = int(rng.choice([1, 2, 13, 64, 200, 255])), int(rng.integers(1, 200))
                data = rng.integers(0, 256, (nb, bl), dtype=np.uint8)
                rows = [(int(rng.integers(0, nb)), int(rng.integers(0, nrep)), int(rng.choice([24, 25, 26, 26])), int(rng.integers(0, bl + 1)),
                         int(rng.integers(0, 16)), 0) for _ in range(nf)]
                rows = [r if r[2] != 26 else r[:4] + (r[4] & 7, 0) for r in rows]
                fl = coast_amd.make_faults(rows)
                exp, est, edet = orc.crc16_xmr(data, bl, replicas=rep, faults=fl, flags=flags)
                det = torch.zeros(nb, dtype=torch.uint8, device="cuda")
                eng.reset_stats()
                eng.inject_faults(fl)
                got = eng.crc16_batch(torch.from_numpy(data.reshape(-1)).cuda(), bl, cfg=coast_amd.XmrConfig(rep, 0, flags), detected=det)
                ok = ((got.cpu().numpy().view(np.uint16) == exp).all() and stats3(eng.stats()) == est and (det.cpu().numpy() == edet).all())
            elif which == "cache_test":
                n, na = int(rng.choice([1, 2, 5, 33, 100])), int(rng.integers(1, 60))
                a = np.tile(np.arange(n, dtype=np.int32), (na, 1))
                if rng.random() < 0.5:
                    hits = rng.integers(0, na * n, int(rng.integers(1, 8)))
                    a.reshape(-1)[hits] = rng.integers(-2**31, 2**31, hits.size, dtype=np.int64).astype(np.int32)
                rows = [(int(rng.integers(0, na)), int(rng.integers(0, nrep)), int(rng.choice([32, 33, 34, 35, 35])), int(rng.integers(0, n + 1)),
                         int(rng.integers(0, 32)), 0) for _ in range(nf)]
                fl = coast_amd.make_faults(rows)
                ea, es, ee, est, edet = orc.cache_test_xmr(a, replicas=rep, faults=fl, flags=flags)
                d = torch.from_numpy(a.copy()).cuda()
                det = torch.zeros(na, dtype=torch.uint8, device="cuda")
                eng.reset_stats()
                eng.inject_faults(fl)
                sums, nerrs = eng.cache_test_batch(d, cfg=coast_amd.XmrConfig(rep, 0, flags), detected=det)
                ok = ((d.cpu().numpy() == ea).all() and (sums.cpu().numpy() == es).all() and (nerrs.cpu().numpy().view(np.uint32) == ee).all()
                      and stats3(eng.stats()) == est and (det.cpu().numpy() == edet).all())
            elif which == "chsha":
                ln, nm = 64 * int(rng.choice([0, 1, 2, 3])), int(rng.integers(1, 40))
                msgs = rng.integers(0, 256, (nm, max(ln, 1)), dtype=np.uint8)
                ncomp = ln // 64 + 1
                rows = []
                for _ in range(nf):
                    site = int(rng.choice([40, 41, 42, 43, 44, 43]))
                    step = (int(rng.integers(0, ncomp)) if site == 42 else int(rng.integers(0, ncomp * 80)) if site < 42
                            else int(rng.integers(0, ncomp * 170)))
                    rows.append((int(rng.integers(0, nm)), int(rng.integers(0, nrep)), site, step, int(rng.integers(0, 32)), int(rng.integers(0, 5))))
                fl = coast_amd.make_faults(rows)
                exp, est, edet = orc.chsha_xmr(msgs, ln, replicas=rep, faults=fl, flags=flags)
                det = torch.zeros(nm, dtype=torch.uint8, device="cuda")
                eng.reset_stats()
                eng.inject_faults(fl)
                got = eng.chsha_batch(torch.from_numpy(msgs).cuda(), ln, cfg=coast_amd.XmrConfig(rep, 0, flags), detected=det)
                ok = ((got.cpu().numpy().view(np.uint32) == exp).all() and stats3(eng.stats()) == est and (det.cpu().numpy() == edet).all())
            else:
                ln, nm = int(rng.choice([0, 1, 3, 55, 56, 64, 65, 119, 120, 130])), int(rng.integers(1, 40))
                if flags & L or (rng.random() < 0.4 and (flags & (B | A)) == (B | A)):  # sha256: the stores' votes belong to the -O0 shape
                    flags |= O0
                msgs = rng.integers(0, 256, (nm, max(ln, 4)), dtype=np.uint8)
                rows = []
                for _ in range(nf):
                    site = int(rng.choice([8, 9, 10, 11, 12, 11, 12]))
                    step = int(rng.integers(0, ln + 2)) if site >= 11 else (int(rng.integers(0, 4)) if site == 10 else int(rng.integers(0, 192)))
                    bit = int(rng.integers(0, 8)) if site >= 11 else int(rng.integers(0, 32))
                    rows.append((int(rng.integers(0, nm)), int(rng.integers(0, nrep)), site, step, bit, int(rng.integers(0, 8))))
                fl = coast_amd.make_faults(rows)
                exp, est, edet = orc.sha256_xmr(msgs, ln, replicas=rep, faults=fl, flags=flags)
                det = torch.zeros(nm, dtype=torch.uint8, device="cuda")
                eng.reset_stats()
                eng.inject_faults(fl)
                got = eng.sha256_batch(torch.from_numpy(msgs).cuda(), ln, cfg=coast_amd.XmrConfig(rep, 0, flags), detected=det).cpu().numpy()
                ok = (got == exp).all() and stats3(eng.stats()) == est and (det.cpu().numpy() == edet).all()
            desc = "walks %s rep=%d flags=%d k=%d" % (which, rep, flags, nf)
        elif kind == "cache_test":
            n = int(rng.choice([1, 2, 3, 4, 5, 31, 32, 33, 36, 64, 100, 128, 600, 601, 1000]))
            na = int(rng.integers(1, 300))
            flags = int(rng.choice([0, 0, 1]))
            a = np.tile(np.arange(n, dtype=np.int32), (na, 1))
            if rng.random() < 0.7:  # memory upsets for the scrub to find
                hits = rng.integers(0, na * n, int(rng.integers(1, 20)))
                a.reshape(-1)[hits] = rng.integers(-2**31, 2**31, hits.size, dtype=np.int64).astype(np.int32)
            hot = rng.integers(0, na, 3)
            fl = faults(rng, int(rng.integers(0, 60)) if rep > 1 else 0, na, nrep, [32, 33, 34], n, 1, hot)
            ea, es, ee, est, edet = orc.cache_test_xmr(a, replicas=rep, faults=fl, flags=flags)
            d = torch.from_numpy(a.copy()).cuda()
            det = torch.zeros(na, dtype=torch.uint8, device="cuda")
            eng.reset_stats()
            eng.inject_faults(fl)
            sums, nerrs = eng.cache_test_batch(d, cfg=coast_amd.XmrConfig(rep, 0, flags), detected=det)
            ok = ((d.cpu().numpy() == ea).all() and (sums.cpu().numpy() == es).all()
                  and (nerrs.cpu().numpy().view(np.uint32) == ee).all() and stats3(eng.stats()) == est
                  and (det.cpu().numpy() == edet).all())
            desc = "cache_test n=%d na=%d rep=%d flags=%d k=%d" % (n, na, rep, flags, len(fl))
        else:
            bl = int(rng.choice([1, 2, 3, 4, 5, 13, 16, 63, 64, 65, 127, 128, 255, 256, 300, 512]))
            nb = int(rng.integers(1, 500))
            sync_every = int(rng.choice([0, 0, 0, 1, 5, 64]))
            data = rng.integers(0, 256, (nb, bl), dtype=np.uint8)
            hot = rng.integers(0, nb, 3)
            fl = faults(rng, int(rng.integers(0, 80)) if rep > 1 else 0, nb, nrep, [24, 25], bl, 1, hot)
            exp, est, edet = orc.crc16_xmr(data, bl, replicas=rep, sync_every=sync_every, faults=fl)
            det = torch.zeros(nb, dtype=torch.uint8, device="cuda")
            eng.reset_stats()
            eng.inject_faults(fl)
            off = int(rng.integers(0, 4)) if rng.random() < 0.3 else 0   # misaligned base pointer
            buf = torch.empty(nb * bl + 16, dtype=torch.uint8, device="cuda")
            buf[off:off + nb * bl].copy_(torch.from_numpy(data.reshape(-1)).cuda())
            got = eng.crc16_batch(buf[off:off + nb * bl], bl, cfg=coast_amd.XmrConfig(rep, sync_every), detected=det)
            got = got.cpu().numpy().view(np.uint16)
            ok = (got == exp).all() and stats3(eng.stats()) == est and (det.cpu().numpy() == edet).all()
            desc = "crc16 bl=%d nb=%d rep=%d V=%d k=%d off=%d" % (bl, nb, rep, sync_every, len(fl), off)
        cases[kind] += 1
        if not ok:
            print("MISMATCH:", desc, "seed", seed, "case", sum(cases.values()))
            sys.exit(1)
    print("fuzz ok:", cases, "in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
