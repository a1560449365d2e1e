#!/usr/bin/env python3
"""campaign.py -- fault-injection campaign on the on-device injector.

Replaces the QEMU/GDB flow of simulation/platform/supervisor.py (one run = boot, pick a uniformly random time and
target, flip one bit of a 32-bit word -- injector.py:202-207, threadFunctions.py:508-520 -- read `C: E: F: T:`) and the
summary of jsonParser.py:148-203.  Here one RUN = one protected work item (a matrix product element's matrix, a
message, an AES block, a CRC block) that receives exactly one single-bit flip at a uniformly random
(replica, site, step, bit); thousands of runs execute as one batch launch.  Classification per run, as
jsonParser.py:162-186 does it:
    error     output differs from the golden (fault-free) output          (E > 0: silent data corruption)
    fault     output correct and a vote saw unequal copies                (F > 0: corrected)
    detected  DWC compare failed (the run would have called FAULT_DETECTED_DWC() and aborted)
    success   output correct, nothing noticed (the flip hit dead state)

    python tools/campaign.py -b mm -m TMR -t 5000          # cf. docs/images/msp430/fault_injection_results2.png
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coast_amd  # noqa: E402

MODES = {"TMR": coast_amd.TMR, "DWC": coast_amd.DWC, "NONE": coast_amd.UNPROTECTED}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-b", "--benchmark", default="mm", choices=["mm", "sha256", "aes", "crc16", "cache_test", "chsha"])
    ap.add_argument("-m", "--mode", default="TMR", choices=list(MODES))
    ap.add_argument("-t", "--runs", type=int, default=5000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--side", type=int, default=9, help="mm: matrix side (one matrix per run)")
    a = ap.parse_args()

    rng = np.random.default_rng(a.seed)
    eng = coast_amd.Engine(0)
    rep = MODES[a.mode]
    nrep = max(rep, 1)
    cfg = coast_amd.XmrConfig(rep)
    clean = coast_amd.XmrConfig(coast_amd.UNPROTECTED)
    g = torch.Generator(device="cuda").manual_seed(a.seed)
    runs = a.runs
    det = torch.zeros(1, dtype=torch.uint8, device="cuda")

    if a.benchmark == "mm":
        n = a.side
        f = torch.randint(-2**31, 2**31, (runs, n, n), dtype=torch.int32, device="cuda", generator=g)
        s = torch.randint(-2**31, 2**31, (runs, n, n), dtype=torch.int32, device="cuda", generator=g)
        gold = eng.mm_batch(f, s, cfg=clean)
        rows = [(run * n * n + int(rng.integers(0, n * n)), int(rng.integers(0, nrep)), int(rng.integers(0, 3)),
                 int(rng.integers(0, n + 1)), int(rng.integers(0, 32))) for run in range(runs)]
        det = torch.zeros(runs * n * n, dtype=torch.uint8, device="cuda")
        eng.reset_stats()
        eng.inject_faults(coast_amd.make_faults(rows))
        out = eng.mm_batch(f, s, cfg=cfg, detected=det)
        bad = (out != gold).reshape(runs, -1).any(dim=1)
        flagged = det.reshape(runs, -1).any(dim=1)
    elif a.benchmark == "sha256":
        msgs = torch.randint(0, 256, (runs, 64), dtype=torch.uint8, device="cuda", generator=g)
        gold = eng.sha256_batch(msgs, 64, cfg=clean)
        rows = []
        for run in range(runs):
            site = int(rng.choice([coast_amd.SITE_SHA_M, coast_amd.SITE_SHA_WV, coast_amd.SITE_SHA_STATE]))
            step = int(rng.integers(0, 3)) if site == coast_amd.SITE_SHA_STATE else int(rng.integers(0, 128))
            rows.append((run, int(rng.integers(0, nrep)), site, step, int(rng.integers(0, 32)), int(rng.integers(0, 8))))
        det = torch.zeros(runs, dtype=torch.uint8, device="cuda")
        eng.reset_stats()
        eng.inject_faults(coast_amd.make_faults(rows))
        out = eng.sha256_batch(msgs, 64, cfg=cfg, detected=det)
        bad = (out != gold).any(dim=1)
        flagged = det.bool()
    elif a.benchmark == "aes":
        st = torch.randint(0, 256, (runs, 16), dtype=torch.uint8, device="cuda", generator=g)
        key = torch.randint(0, 256, (runs, 16), dtype=torch.uint8, device="cuda", generator=g)
        gs, gk = st.clone(), key.clone()
        eng.aes128_batch(gs, gk, 0, cfg=clean)
        rows = [(run, int(rng.integers(0, nrep)), int(rng.choice([coast_amd.SITE_AES_STATE, coast_amd.SITE_AES_KEY])),
                 int(rng.integers(0, 11)), int(rng.integers(0, 32)), int(rng.integers(0, 4))) for run in range(runs)]
        det = torch.zeros(runs, dtype=torch.uint8, device="cuda")
        eng.reset_stats()
        eng.inject_faults(coast_amd.make_faults(rows))
        eng.aes128_batch(st, key, 0, cfg=cfg, detected=det)
        bad = (st != gs).any(dim=1) | (key != gk).any(dim=1)
        flagged = det.bool()
    elif a.benchmark == "chsha":
        ln = 192  # three data blocks + the padding block per run
        msgs = torch.randint(0, 256, (runs, ln), dtype=torch.uint8, device="cuda", generator=g)
        gold = eng.chsha_batch(msgs, ln, cfg=clean)
        rows = []
        for run in range(runs):
            site = int(rng.choice([coast_amd.SITE_CHSHA_W, coast_amd.SITE_CHSHA_WV, coast_amd.SITE_CHSHA_DIGEST]))
            step = int(rng.integers(0, 4)) if site == coast_amd.SITE_CHSHA_DIGEST else int(rng.integers(0, 4 * 80))
            rows.append((run, int(rng.integers(0, nrep)), site, step, int(rng.integers(0, 32)), int(rng.integers(0, 5))))
        det = torch.zeros(runs, dtype=torch.uint8, device="cuda")
        eng.reset_stats()
        eng.inject_faults(coast_amd.make_faults(rows))
        out = eng.chsha_batch(msgs, ln, cfg=cfg, detected=det)
        bad = (out != gold).any(dim=1)
        flagged = det.bool()
    elif a.benchmark == "cache_test":
        n = 600  # data_array_elements, cacheTest.c:78
        arr = torch.arange(n, dtype=torch.int32, device="cuda").repeat(runs, 1).contiguous()
        gs, ge = eng.cache_test_batch(arr.clone(), cfg=clean)
        rows = [(run, int(rng.integers(0, nrep)), int(rng.choice([coast_amd.SITE_CT_SUM, coast_amd.SITE_CT_VAL,
                                                                  coast_amd.SITE_CT_NERR])),
                 int(rng.integers(0, n + 1)), int(rng.integers(0, 32))) for run in range(runs)]
        det = torch.zeros(runs, dtype=torch.uint8, device="cuda")
        eng.reset_stats()
        eng.inject_faults(coast_amd.make_faults(rows))
        work = arr.clone()
        sums, nerrs = eng.cache_test_batch(work, cfg=cfg, detected=det)
        bad = (sums != gs) | (nerrs != ge) | (work != arr).any(dim=1)
        flagged = det.bool()
    else:
        bl = 255  # the reference's maximum length (unsigned char, crc16.c:21)
        data = torch.randint(0, 256, (runs * bl,), dtype=torch.uint8, device="cuda", generator=g)
        gold = eng.crc16_batch(data, bl, cfg=clean)
        rows = [(run, int(rng.integers(0, nrep)), int(rng.choice([coast_amd.SITE_CRC_CRC, coast_amd.SITE_CRC_X])),
                 int(rng.integers(0, bl + 1)), int(rng.integers(0, 32))) for run in range(runs)]
        det = torch.zeros(runs, dtype=torch.uint8, device="cuda")
        eng.reset_stats()
        eng.inject_faults(coast_amd.make_faults(rows))
        out = eng.crc16_batch(data, bl, cfg=cfg, detected=det)
        bad = out != gold
        flagged = det.bool()

    st_ = eng.stats()
    bad, flagged = bad.cpu().numpy(), flagged.cpu().numpy()
    if rep == coast_amd.DWC:
        detected = int(flagged.sum())
        errors = int((bad & ~flagged).sum())
        faults = 0
    else:
        detected = 0
        errors = int(bad.sum())
        faults = int((flagged & ~bad).sum())
    success = runs - errors - faults - detected
    out = {"benchmark": a.benchmark, "mode": a.mode, "runs": runs, "success": success, "faults_corrected": faults,
           "errors_sdc": errors, "dwc_detected": detected, "coverage_pct": 100.0 * (runs - errors) / runs,
           "TMR_ERROR_CNT": st_["errors_corrected"], "__SYNC_COUNT": st_["sync_count"],
           "fault_model": "one single-bit flip of a replica-private 32-bit register per run (injector.py:202-207)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
