#!/usr/bin/env python3
"""aes_fold_check.py -- the opt-in counter fold inside the persistent aes kernels (COAST_AES_FOLD=1, block_fold in xmr.hpp) against the
separate fold kernel: same states, keys and totals -- errors, syncs, detected items, LAUNCHES -- over steps that alternate directions, with
launches of another kernel (whose counts sit in the slots when the aes kernel folds) in between, with and without an explicit
reduce_counters after every launch, the ticket back at zero for the next launch every time.  A process of its own
(tests/test_gpu_parity.py::test_aes_persistent_kernels_fold_their_own_counters starts it): the path is opt-in because a GPU memory fault on
it is still unexplained, and a fault must not take the suite's process down.  Usage: tests/aes_fold_check.py [replicas]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coast_amd  # noqa: E402


def rand_faults(rng, k, nitems, nrep):
    rows = [(int(rng.integers(0, nitems)), int(rng.integers(0, nrep)), int(rng.choice([16, 17])), int(rng.integers(0, 11)),
             int(rng.integers(0, 32)), int(rng.integers(0, 4))) for _ in range(k)]
    return coast_amd.make_faults(rows)


def main():
    replicas = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    eng = coast_amd.Engine(0)
    rng = np.random.default_rng(4242 + replicas)
    n = 70001
    st0 = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    key0 = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    data = torch.from_numpy(rng.integers(0, 256, (4096, 64), dtype=np.uint8)).cuda()
    os.environ["COAST_AES_TABLES"] = "replicated"  # (read by the library at every call)
    got = {}
    for fold in ("1", "0"):
        os.environ["COAST_AES_FOLD"] = fold
        for explicit in (False, True):
            ds, dk = torch.from_numpy(st0.copy()).cuda(), torch.from_numpy(key0.copy()).cuda()
            eng.reset_stats()
            seen = []
            for step in range(6):
                fl = rand_faults(np.random.default_rng(step), 40, n, replicas)
                if step % 3 == 2:  # another kernel's counts in the slots, its launch pending
                    eng.crc16_batch(data, 64, cfg=coast_amd.XmrConfig(3))
                eng.inject_faults(fl)
                eng.aes128_batch(ds, dk, step & 1, cfg=coast_amd.XmrConfig(replicas))
                if explicit:
                    eng.reduce_counters()
                if step % 2:
                    st = eng.stats()
                    seen.append(tuple(st[k] for k in ("errors_corrected", "sync_count", "dwc_detected", "launches")))
            got[(fold, explicit)] = (ds.cpu().numpy(), dk.cpu().numpy(), seen)
    ref = got[("0", True)]
    assert ref[2][-1][-1] == 8 and ref[2][0][0] + ref[2][0][2] > 0, ref[2]
    for k, v in got.items():
        assert (v[0] == ref[0]).all() and (v[1] == ref[1]).all() and v[2] == ref[2], (k, v[2], ref[2])
    print("aes fold ok: replicas %d, totals %s" % (replicas, ref[2][-1]))


if __name__ == "__main__":
    main()
