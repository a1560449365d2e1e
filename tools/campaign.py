#!/usr/bin/env python3
"""campaign.py -- fault-injection campaign front-end on the MI355X engine.

Replaces the QEMU/GDB flow of simulation/platform/supervisor.py: there one run = boot the benchmark, stop it at a uniformly
random time, flip one bit of a 32-bit word in a random register or in a chosen memory section (resources/injector.py:202-260,
threadFunctions.py:508-520), let it finish and read `C: E: F: T:` from the UART; the campaign is logged per run
(supervisor.py:420-437: <board>_<benchmark>_<timestamp>.log / .json) and summarised by jsonParser.py:148-203.  Here one RUN =
one protected work item (one matrix product, one message, one AES block, one CRC block ...) that receives exactly one
single-bit upset; thousands of runs execute as ONE batch launch, and the same artefacts come out: a per-run .json, the UART
lines in a .log, and the jsonParser summary table.

    -s registers   a replica-private register (the on-device injector: uniformly random replica / site / step / bit)
    -s memory      a random bit of a random 32-bit word of the run's input memory (coast_flip_memory = injectFaultMem)
    --mem-mode nomemrep   the lane-replicated engine: `-TMR -noMemReplication` -- one memory copy, so a memory upset reaches
                          every replica (reference: 86.3 % coverage ~ unmitigated, docs/images/msp430/fault_injection_results.png)
    --mem-mode default    COAST's default mode: one unprotected launch per memory copy + the exit vote (coast_sync_copies);
                          the upset sits in one copy and is out-voted (reference: 98.8 %)
    --mem-mode storesync  memory replicated + -storeDataSync: one launch of the lane-replicated kernel on `replicas` memory
                          copies (COAST_F_MEMORY_COPIES: sha256, aes, crc16); the upset is out-voted at the next store

Classification per run, exactly jsonParser.summarizeRuns (:162-186): errors > 0 -> error; else faults > 0 -> fault (counted
with the successes in "Successes"); a DWC compare failure is FAULT_DETECTED_DWC() -> abort(), which the reference's
supervisor sees as abort + timeout.

    python tools/campaign.py -b mm -m TMR -t 5000 --side 256            # register upsets, matrix-core engine
    python tools/campaign.py -b crc16 -m TMR -t 5000 -s memory --mem-mode default
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coast_amd as ca  # noqa: E402

def cu_count():
    """compute units of the device (the matrix-core kernel runs one workgroup per CU, four per matrix); 256 = an MI355X, for the CPU-side tests"""
    return torch.cuda.get_device_properties(0).multi_processor_count if torch.cuda.is_available() else 256


MODES = {"TMR": ca.TMR, "DWC": ca.DWC, "NONE": ca.UNPROTECTED, "CFCSS": ca.UNPROTECTED}  # CFCSS: -b crazycf only


# ------------------------------------------------------------------------------------------------ benchmarks
class Bench:
    """inputs(): fresh device tensors of `runs` work items; run(): one protected launch -> (runs, k) output tensor;
    memory(): the input tensors that make up a run's memory image; reg_fault(): one random register upset of run r."""


class MM(Bench):
    def __init__(self, a, eng, g):
        self.n, self.eng = a.side, eng

    def inputs(self, runs, g):
        n = self.n
        return [torch.randint(-2**31, 2**31, (runs, n, n), dtype=torch.int32, device="cuda", generator=g) for _ in range(2)]

    def run(self, inp, cfg, det=None):
        d = None if det is None else torch.zeros(inp[0].numel(), dtype=torch.uint8, device="cuda")
        out = self.eng.mm_batch(inp[0], inp[1], cfg=cfg, detected=d)
        if det is not None:
            det |= d.reshape(inp[0].shape[0], -1).any(dim=1).to(torch.uint8)
        return out.reshape(inp[0].shape[0], -1)

    def reg_fault(self, r, nrep, rng):
        n = self.n
        return (r * n * n + int(rng.integers(0, n * n)), int(rng.integers(0, nrep)), int(rng.integers(0, 3)),
                int(rng.integers(0, n + 1)), int(rng.integers(0, 32)))

    def counter_fault(self, r, nrep, rng):  # --counters-in-sor: i / j / k / sum of the call, before loop condition `step`
        n = self.n
        return (r * n * n, int(rng.integers(0, nrep)), int(rng.choice([ca.SITE_MM_I, ca.SITE_MM_J, ca.SITE_MM_K, ca.SITE_MM_ACC])),
                int(rng.integers(0, (n + 1) * (n * n + n + 1))), int(rng.integers(0, 32)))

    # Register census of one wave of the TMR matrix-core kernel at side 256: 256 VGPRs x 64 lanes x 32 bits, named in the kernel
    # source (tests/test_kernel_budget_cpu.py holds the total).  `getReg()` of the reference draws uniformly from the register class
    # (simulation/platform/resources/injector.py:70-72, 237-260); here one draw = one bit of one lane of one VGPR.  SGPRs are 32 bits
    # per wave against 2048 per VGPR: < 1 % of the bits, filed under "other".
    #   name      regs  what one flipped bit reaches
    CENSUS_BLK3 = [  # mm_mfma_blk3_kernel<3> (the default since round 4): every loaded operand is replica-private
        ("acc",     96, "private: limb sum C_t of one replica of one output element (2 row blocks x 3 replicas x 4 limbs x 4)"),
        ("b_frag",  48, "private: 16 plane bytes of s[k..k+15][j] in ONE replica's B-operand registers: that replica of the wave's 32 rows"),
        ("a_frag",  16, "private: 4 plane bytes of f[i][k..k+3] in the A-operand registers of ONE replica's set of ten MFMAs (re-read from "
                        "the LDS panel per replica): that replica of the tile's 16 columns"),
        ("s_raw",   24, "COMMON: a raw / half-converted word of s[k][j] on its way into the LDS slab both waves of the pair and all "
                        "three replicas read: the panel's 64 rows of column j"),
        ("f_raw",    4, "COMMON: a raw word of f[i][k] of the next panel on its way into the LDS panel: all 256 columns of row i"),
        ("tally",    8, "vote scratch and counters (teV, agree, nExec, nReal): the stored words are not reached"),
        ("other",   60, "addresses, lane constants, compiler temporaries (+ the wave's SGPRs): not modelled"),
    ]
    CENSUS_BLK2 = [  # mm_mfma_blk2_kernel<3> (COAST_MM_TILE=blocks2, the default of rounds 2-3): ONE A fragment set for the three replicas
        ("acc",     96, "private"), ("b_frag", 48, "private"),
        ("a_frag",  16, "COMMON: 4 plane bytes of f[i][k..k+3], the shared A operand: all three replicas of the tile's 16 columns"),
        ("s_raw",   24, "COMMON"), ("f_raw", 4, "COMMON"), ("tally", 8, ""), ("other", 60, ""),
    ]

    @classmethod
    def census(cls):
        return cls.CENSUS_BLK2 if os.environ.get("COAST_MM_TILE") == "blocks2" else cls.CENSUS_BLK3

    def reg_event(self, r, nrep, rng):
        """one physical register upset of run r (= matrix r) on the matrix-core TMR kernel -> (class, rows for the injector).
        A limb-sum / plane-byte bit is mapped onto the model's 32-bit registers: limb t bit b = bit 8 t + b of the word (beyond
        bit 31: no architectural effect); the operand sites take the bit of the operand value."""
        n, nn = self.n, self.n * self.n
        census = self.census()
        w = np.array([c[1] for c in census], dtype=np.float64)
        cls = census[int(rng.choice(len(w), p=w / w.sum()))][0]
        i, j, k = int(rng.integers(0, n)), int(rng.integers(0, n)), int(rng.integers(0, n))
        limb, b = int(rng.integers(0, 4)), int(rng.integers(0, 32 if cls == "acc" else 8))
        bit = 8 * limb + b
        item = lambda ii, jj: r * nn + ii * n + jj
        if cls in ("tally", "other"):
            return cls, []
        if getattr(self, "real_staging", False) and cls in ("s_raw", "f_raw"):
            # --reg-model physical-real-all: the shared staging registers as real flips too: raw words of s (16 VGPRs: registers 12-19 x 2 dwords) / of the next f panel
            # (register 20 x 4 dwords); the conversion temporaries of the census' 24 are dead between stages (a flip there: no effect)
            if cls == "s_raw" and rng.random() < 8.0 / 24.0:
                return cls, []
            reg, dw = (20, int(rng.integers(0, 4))) if cls == "f_raw" else (12 + int(rng.integers(0, 8)), int(rng.integers(0, 2)))
            step = int(rng.integers(0, 4)) | (int(rng.integers(0, 64)) << 8) | (dw << 16) | (reg << 24)
            return cls, [(item(i, j), int(rng.integers(0, nrep)), ca.SITE_MM_VGPR, step, int(rng.integers(0, 32)))]
        if getattr(self, "real", False) and cls in ("acc", "b_frag", "a_frag"):
            # --reg-model physical-real: the replica-private classes are REAL flips (COAST_SITE_MM_VGPR: an exclusive-or on the named
            # register of the running kernel) -- any lane, any dword of the fragment, any of the 32 bits; the outcome is the hardware's
            reg = {"a_frag": 0, "b_frag": 4, "acc": 8}[cls] + limb
            slab = int(rng.integers(1, 4)) if cls == "acc" else int(rng.integers(0, 4))
            step = slab | (int(rng.integers(0, 64)) << 8) | (int(rng.integers(0, 4)) << 16) | (reg << 24)
            return cls, [(item(i, j), int(rng.integers(0, nrep)), ca.SITE_MM_VGPR, step, int(rng.integers(0, 32)))]
        if cls == "acc":
            if bit > 31:  # a bit of a limb sum that falls off the 32-bit word: no architectural effect
                return cls, []
            return cls, [(item(i, j), int(rng.integers(0, nrep)), ca.SITE_MM_ACC, int(rng.integers(0, n + 1)), bit)]
        if cls == "b_frag":  # one replica's copy of s[k][j]: the 32 rows (two row blocks) the wave's accumulators stand for
            rep, i0 = int(rng.integers(0, nrep)), (i // 32) * 32
            return cls, [(item(ii, j), rep, ca.SITE_MM_OPB, k, bit) for ii in range(i0, i0 + 32)]
        if cls == "a_frag":  # the 16 columns of the tile; blocks3: one replica's set of MFMAs, blocks2: shared by the three replicas
            j0 = (j // 16) * 16
            rep = ca.REPLICA_ALL if census is self.CENSUS_BLK2 else int(rng.integers(0, nrep))
            return cls, [(item(i, jj), rep, ca.SITE_MM_OPA, k, bit) for jj in range(j0, j0 + 16)]
        if cls == "s_raw":   # shared by the three replicas and the 64 rows of the panel
            i0 = (i // 64) * 64
            return cls, [(item(ii, j), ca.REPLICA_ALL, ca.SITE_MM_OPB, k, bit) for ii in range(i0, i0 + 64)]
        return cls, [(item(i, jj), ca.REPLICA_ALL, ca.SITE_MM_OPA, k, bit) for jj in range(n)]  # f_raw


class SHA256(Bench):
    def __init__(self, a, eng, g):
        self.eng, self.len = eng, 64

    def inputs(self, runs, g):
        return [torch.randint(0, 256, (runs, self.len), dtype=torch.uint8, device="cuda", generator=g)]

    def run(self, inp, cfg, det=None):
        return self.eng.sha256_batch(inp[0], self.len, cfg=cfg, detected=det)

    def reg_fault(self, r, nrep, rng):
        site = int(rng.choice([ca.SITE_SHA_M, ca.SITE_SHA_WV, ca.SITE_SHA_STATE]))
        step = int(rng.integers(0, 3)) if site == ca.SITE_SHA_STATE else int(rng.integers(0, 128))
        return (r, int(rng.integers(0, nrep)), site, step, int(rng.integers(0, 32)), int(rng.integers(0, 8)))

    def counter_fault(self, r, nrep, rng):  # the byte loop's i / ctx_datalen before byte-loop iteration `step`
        return (r, int(rng.integers(0, nrep)), int(rng.choice([ca.SITE_SHA_I, ca.SITE_SHA_DATALEN])), int(rng.integers(0, self.len + 1)),
                int(rng.integers(0, 32)))


class AES(Bench):
    def __init__(self, a, eng, g):
        self.eng = eng

    def inputs(self, runs, g):
        return [torch.randint(0, 256, (runs, 16), dtype=torch.uint8, device="cuda", generator=g) for _ in range(2)]

    def run(self, inp, cfg, det=None):
        st, key = inp[0].clone(), inp[1].clone()  # aes_enc_dec works in place on state AND key (TI_aes_128.c:107-231)
        self.eng.aes128_batch(st, key, 0, cfg=cfg, detected=det)
        return torch.cat([st, key], dim=1)

    def reg_fault(self, r, nrep, rng):
        return (r, int(rng.integers(0, nrep)), int(rng.choice([ca.SITE_AES_STATE, ca.SITE_AES_KEY])),
                int(rng.integers(0, 11)), int(rng.integers(0, 32)), int(rng.integers(0, 4)))

    def counter_fault(self, r, nrep, rng):  # round / i (8 bits live) before loop condition `step` of the encryption walk (373 of them)
        return (r, int(rng.integers(0, nrep)), int(rng.choice([ca.SITE_AES_ROUND, ca.SITE_AES_I])), int(rng.integers(0, 373)),
                int(rng.integers(0, 8)))


class CRC16(Bench):
    def __init__(self, a, eng, g):
        self.eng, self.bl = eng, 255  # the reference's maximum length (unsigned char, crc16.c:21) ... rows padded to 256

    def inputs(self, runs, g):
        return [torch.randint(0, 256, (runs, 256), dtype=torch.uint8, device="cuda", generator=g)]

    def run(self, inp, cfg, det=None):
        data = inp[0][:, : self.bl].contiguous().reshape(-1)
        return self.eng.crc16_batch(data, self.bl, cfg=cfg, detected=det).reshape(-1, 1)

    def reg_fault(self, r, nrep, rng):
        return (r, int(rng.integers(0, nrep)), int(rng.choice([ca.SITE_CRC_CRC, ca.SITE_CRC_X])),
                int(rng.integers(0, self.bl + 1)), int(rng.integers(0, 32)))

    def counter_fault(self, r, nrep, rng):  # `length` (8 bits live) before the loop condition of iteration `step`
        return (r, int(rng.integers(0, nrep)), ca.SITE_CRC_LEN, int(rng.integers(0, self.bl + 1)), int(rng.integers(0, 8)))


class ChSha(Bench):
    def __init__(self, a, eng, g):
        self.eng, self.len = eng, 192

    def inputs(self, runs, g):
        return [torch.randint(0, 256, (runs, self.len), dtype=torch.uint8, device="cuda", generator=g)]

    def run(self, inp, cfg, det=None):
        return self.eng.chsha_batch(inp[0], self.len, cfg=cfg, detected=det)

    def reg_fault(self, r, nrep, rng):
        site = int(rng.choice([ca.SITE_CHSHA_W, ca.SITE_CHSHA_WV, ca.SITE_CHSHA_DIGEST]))
        step = int(rng.integers(0, 4)) if site == ca.SITE_CHSHA_DIGEST else int(rng.integers(0, 4 * 80))
        return (r, int(rng.integers(0, nrep)), site, step, int(rng.integers(0, 32)), int(rng.integers(0, 5)))

    def counter_fault(self, r, nrep, rng):  # sha_transform's i / sha_update's count before loop condition `step` (4 transforms x 167)
        return (r, int(rng.integers(0, nrep)), int(rng.choice([ca.SITE_CHSHA_I, ca.SITE_CHSHA_I, ca.SITE_CHSHA_COUNT])),
                int(rng.integers(0, 4 * 167)), int(rng.integers(0, 32)))


class CacheTest(Bench):
    def __init__(self, a, eng, g):
        self.eng, self.n = eng, 600  # data_array_elements, cacheTest.c:78

    def inputs(self, runs, g):
        return [torch.arange(self.n, dtype=torch.int32, device="cuda").repeat(runs, 1).contiguous()]

    def run(self, inp, cfg, det=None):
        work = inp[0].clone()  # calc_sum scrubs the array in place
        sums, nerrs = self.eng.cache_test_batch(work, cfg=cfg, detected=det)
        return torch.cat([sums.reshape(-1, 1), nerrs.reshape(-1, 1), work], dim=1)

    def reg_fault(self, r, nrep, rng):
        return (r, int(rng.integers(0, nrep)), int(rng.choice([ca.SITE_CT_SUM, ca.SITE_CT_VAL, ca.SITE_CT_NERR])),
                int(rng.integers(0, self.n + 1)), int(rng.integers(0, 32)))

    def counter_fault(self, r, nrep, rng):  # calc_sum's i before loop condition `step`
        return (r, int(rng.integers(0, nrep)), ca.SITE_CT_I, int(rng.integers(0, self.n + 1)), int(rng.integers(0, 32)))


class QuickSort(Bench):
    def __init__(self, a, eng, g):
        self.eng, self.n = eng, 580  # array_elements, tests/quicksort/quicksort.c:82
        self.status = None

    def inputs(self, runs, g):
        return [torch.randint(-2**31, 2**31, (runs, self.n), dtype=torch.int32, device="cuda", generator=g)]

    def run(self, inp, cfg, det=None):
        work = inp[0].clone()  # quick_sort works in place
        self.status = torch.zeros(work.shape[0], dtype=torch.uint8, device="cuda")
        self.eng.quicksort_batch(work, cfg=cfg, detected=det, status=self.status)
        return work

    def reg_fault(self, r, nrep, rng):
        return (r, int(rng.integers(0, nrep)), int(rng.integers(ca.SITE_QS_I, ca.SITE_QS_VJ + 1)), int(rng.integers(0, 12000)),
                int(rng.integers(0, 32)))


class ChAes(Bench):
    """CHStone aes (tests/chstone/aes): one Rijndael block per run; --chaes-type picks one of the nine key / block sizes"""

    def __init__(self, a, eng, g):
        self.eng, self.type = eng, a.chaes_type
        self.nk, self.nb = self.type // 1000 // 32, self.type % 1000 // 32
        self.nr = max(self.nk, self.nb) + 6

    def inputs(self, runs, g):
        return [torch.randint(0, 256, (runs, 4 * self.nb), dtype=torch.uint8, device="cuda", generator=g),
                torch.randint(0, 256, (runs, 4 * self.nk), dtype=torch.uint8, device="cuda", generator=g)]

    def run(self, inp, cfg, det=None):
        st = inp[0].clone()
        self.eng.chaes_batch(st, inp[1], self.type, 0, cfg, detected=det)
        return st

    def reg_fault(self, r, nrep, rng):
        if rng.random() < 0.5:
            return (r, int(rng.integers(0, nrep)), ca.SITE_CHAES_STATE, int(rng.integers(0, self.nr + 2)), int(rng.integers(0, 32)),
                    int(rng.integers(0, self.nb)))
        return (r, int(rng.integers(0, nrep)), ca.SITE_CHAES_WORD, int(rng.integers(0, self.nb * (self.nr + 1))),
                int(rng.integers(0, 32)))

    def counter_fault(self, r, nrep, rng):  # encrypt's round counter / the callees' j, i before loop condition `step` of the encryption walk
        nk, nb, nr = self.nk, self.nb, self.nr
        cols = nb * (nr + 1)
        ks = ((nk + 1) + 5 * nk) + ((cols - nk + 1) + 5 * (cols - nk) + (5 * ((cols - nk) // nk) if nk > 6 else 0))  # KeySchedule's two loops
        ticks = ks + 2 * (nb + 1) + nr + (nr - 1) * 2 * (nb + 1)  # + two AddRoundKey calls, the round loop, MixColumn's two loops
        step = int(rng.integers(0, ticks))
        # aim at a counter that is live at that point: KeySchedule runs on j (i only inside its inner loops), the rounds on the round
        # counter and the callees' j
        site = ca.SITE_CHAES_J if step < ks and rng.random() < 0.8 else ca.SITE_CHAES_I if step < ks else int(rng.choice([ca.SITE_CHAES_RND, ca.SITE_CHAES_J]))
        return (r, int(rng.integers(0, nrep)), site, step, int(rng.integers(0, 32)))


class CrazyCF(Bench):
    """tests/crazyCF/crazyCF.c under `opt -CFCSS` (-m CFCSS) or bare (-m NONE): the upset is a corrupted branch target -- execution
    lands at the start of block (target ^ 1 << bit), bit uniform over the 32-bit register -- or, under CFCSS, a bit of the two
    signature globals.  A failed signature check is FAULT_DETECTED_CFC() -> abort(); a jump that leaves the program, or a run
    the watchdog cuts, is what the supervisor files under timeouts."""

    def __init__(self, a, eng, g):
        self.eng, self.cfcss = eng, a.mode == "CFCSS"
        self.status = None
        # -m TMR / DWC (unittest/cfg/full_tmr.yml:8 runs the program with -TMR), or -m NONE --counters-in-sor for the same upsets on the
        # unprotected run: lane-replicated runs of main() (coast_crazycf_xmr_batch), the upset is a bit of i / total / timesThroughWhile /
        # fillArray's i before one of the run's 82 branch conditions
        self.xmr = a.mode in ("TMR", "DWC") or bool(getattr(a, "counters_in_sor", False))

    def inputs(self, runs, g):
        prm = torch.tensor([[42, 20, 10]], dtype=torch.int32, device="cuda").repeat(runs, 1)  # crazyCF.c:36, 11, 41
        return [prm]

    def run(self, inp, cfg, det=None):
        if self.xmr:
            if cfg.replicas > 1:  # the replicated registers ARE the program's counters: their branch / switch / offset votes belong to the mode
                cfg = ca.XmrConfig(cfg.replicas, 0, cfg.flags | ca.F_BRANCH_SYNC | ca.F_ADDR_SYNC)
            res, st = self.eng.crazycf_xmr_batch(inp[0], cfg, detected=det)
            self.status = (st == ca.CFC_WATCHDOG).to(torch.uint8)
            return res[:, :3].contiguous()
        res, st = self.eng.crazycf_batch(inp[0], cfcss=self.cfcss)
        if det is not None:
            det.copy_((st == ca.CFC_DETECTED).to(torch.uint8))
            self.status = ((st == ca.CFC_WATCHDOG) | (st == ca.CFC_WILD)).to(torch.uint8)
        return res[:, :3].contiguous()  # Total, the "total so far" value, how many such lines

    def counter_fault(self, r, nrep, rng):
        return (r, int(rng.integers(0, nrep)), int(rng.choice([ca.SITE_CCF_I, ca.SITE_CCF_TOTAL, ca.SITE_CCF_TIMES, ca.SITE_CCF_FI])),
                int(rng.integers(0, 82)), int(rng.integers(0, 32)))

    def reg_fault(self, r, nrep, rng):
        if self.xmr:
            return self.counter_fault(r, nrep, rng)
        site = ca.SITE_CFC_PC
        if self.cfcss and rng.random() < 0.4:
            site = ca.SITE_CFC_RTS if rng.random() < 0.5 else ca.SITE_CFC_RTSA
        return (r, 0, site, int(rng.integers(0, 309)), int(rng.integers(0, 32)))  # 309 block transitions in a clean run


BENCHES = {"chaes": ChAes, "crazycf": CrazyCF, "quicksort": QuickSort, "mm": MM, "sha256": SHA256, "aes": AES, "crc16": CRC16, "chsha": ChSha, "cache_test": CacheTest}


# ------------------------------------------------------------------------------------------------ one campaign
def flip_memory(eng, inp, r, rng):
    """injectFaultMem (injector.py:209-235) on run r's memory image: a uniformly random 32-bit word, a uniformly random bit"""
    sizes = [t[r].numel() * t.element_size() for t in inp]
    words = [s // 4 for s in sizes]
    w = int(rng.integers(0, sum(words)))
    k = 0
    while w >= words[k]:
        w -= words[k]
        k += 1
    bit = int(rng.integers(0, 32))
    byte_off = r * sizes[k] + 4 * w + bit // 8
    eng.flip_memory(inp[k], byte_off, bit % 8)
    return {"tensor": k, "word": w, "bit": bit}


def run_campaign(a, eng=None):
    rng = np.random.default_rng(a.seed)
    eng = eng or ca.Engine(0)
    g = torch.Generator(device="cuda").manual_seed(a.seed)
    rep = MODES[a.mode]
    if (a.mode == "CFCSS") != (a.benchmark == "crazycf" and a.mode == "CFCSS"):
        raise SystemExit("-m CFCSS applies to -b crazycf (the reference's CFCSS test program)")
    if a.benchmark == "crazycf" and a.section != "registers":
        raise SystemExit("-b crazycf: -s registers")
    aborting = rep == ca.DWC or a.mode == "CFCSS"  # a detection calls the handler -> abort()
    nrep = max(rep, 1)
    runs = a.runs
    bench = BENCHES[a.benchmark](a, eng, g)
    inp = bench.inputs(runs, g)
    clean = ca.XmrConfig(ca.UNPROTECTED)
    gold = bench.run(inp, clean).clone()
    det = torch.zeros(runs, dtype=torch.uint8, device="cuda")
    targets = []
    t0 = time.perf_counter()
    eng.reset_stats()
    classes = bad_staged = None
    if a.section == "registers":
        physical = a.reg_model.startswith("physical") and a.benchmark == "mm" and a.side == 256 and rep == ca.TMR
        if a.reg_model.startswith("physical") and not physical:
            raise SystemExit("--reg-model physical: the register census is that of the TMR matrix-core kernel (-b mm --side 256 -m TMR)")
        bench.real = a.reg_model in ("physical-real", "physical-real-all")
        bench.real_staging = a.reg_model == "physical-real-all"
        if physical:  # any register of the wave, weighted by the kernel's register census: shared state included
            rows, classes, staged_rows = [], [], []
            for r in range(runs):
                cls, ev = bench.reg_event(r, nrep, rng)
                classes.append(cls)
                if bench.real_staging and cls in ("s_raw", "f_raw"):
                    staged_rows += ev  # physical-real-all: the staging flips run in a launch of their own (below)
                else:
                    rows += ev
                targets.append({"class": cls, "flips": len(ev), "site": ev[0][2] if ev else None, "step": ev[0][3] if ev else None,
                                "bit": ev[0][4] if ev else None, "replica": ev[0][1] if ev else None})
        else:
            if a.counters_in_sor:  # the loop counters are members of the sphere of replication and the campaign aims at THEM
                if not hasattr(bench, "counter_fault") or (a.benchmark == "mm" and a.side > 32):
                    raise SystemExit("--counters-in-sor: mm (--side <= 32), sha256, aes, crc16, chsha, chaes, crazycf, cache_test")
                rows = [bench.counter_fault(r, nrep, rng) for r in range(runs)]
            else:
                rows = [bench.reg_fault(r, nrep, rng) for r in range(runs)]
            for r, row in enumerate(rows):
                targets.append({"replica": row[1], "site": row[2], "step": row[3], "bit": row[4]})
        eng.inject_faults(ca.make_faults(rows))
        flags = (ca.F_BRANCH_SYNC if a.benchmark == "crc16" else ca.F_BRANCH_SYNC | ca.F_ADDR_SYNC) if a.counters_in_sor else 0
        if a.benchmark == "mm" and not getattr(a, "clone_staging", False):
            flags |= ca.F_SINGLE_STAGING  # (the library clones the staging loads by default since ABI 8; the campaign's rows name the form they ran)
        out = bench.run(inp, ca.XmrConfig(rep, 0, flags), det)
        engine = eng.last_launch()
        if physical and bench.real_staging:
            # a REAL flip of a staging register can hold a word of the workgroup's NEXT matrix (its slabs and its f panel are staged
            # ahead): the wrong words then belong to run r + stride.  A second launch with only the staging flips armed keeps those
            # errors apart from the outcome of the other runs' own upsets (first launch) -- exact, no guess about who corrupted what.
            bad_staged = np.zeros(runs, dtype=bool)
            if staged_rows:
                eng.inject_faults(ca.make_faults(staged_rows))
                out2 = bench.run(inp, ca.XmrConfig(rep, 0, flags), det)
                bad_staged = (out2.reshape(runs, -1) != gold.reshape(runs, -1)).any(dim=1).cpu().numpy()
    elif a.mem_mode == "nomemrep" or rep == ca.UNPROTECTED:
        for r in range(runs):  # the single memory copy is hit: every replica loads the same corrupted word
            targets.append(flip_memory(eng, inp, r, rng))
        out = bench.run(inp, ca.XmrConfig(rep), det)
        engine = eng.last_launch()
    elif a.mem_mode == "storesync":
        # the memory-replicated mode with -storeDataSync (dataflowProtection.cpp:14-18): ONE launch of the lane-replicated lean
        # kernel on `replicas` memory copies (COAST_F_MEMORY_COPIES) -- replica r loads from copy r, stores are voted into every copy
        if a.benchmark not in ("sha256", "aes", "crc16"):
            raise SystemExit("--mem-mode storesync: sha256, aes, crc16 (the kernels that implement COAST_F_MEMORY_COPIES)")
        stacked = [torch.stack([t] * nrep).contiguous() for t in inp]
        for r in range(runs):
            k = int(rng.integers(0, nrep))
            tgt = flip_memory(eng, [t[k] for t in stacked], r, rng)
            tgt["copy"] = k
            targets.append(tgt)
        cfgc = ca.XmrConfig(rep, 0, ca.F_MEMORY_COPIES)
        if a.benchmark == "sha256":
            o = eng.sha256_batch(stacked[0], bench.len, cfg=cfgc, detected=det)
        elif a.benchmark == "aes":
            st, key = stacked[0].clone(), stacked[1].clone()
            eng.aes128_batch(st, key, 0, cfg=cfgc, detected=det)
            o = torch.cat([st, key], dim=2)
        else:
            o = eng.crc16_batch(stacked[0][:, :, : bench.bl].contiguous().reshape(nrep, -1), bench.bl, cfg=cfgc, detected=det).reshape(nrep, runs, 1)
        out = o[0].reshape(runs, -1)  # (under TMR every copy holds the voted result)
        engine = eng.last_launch()
    else:
        # default mode (docs/source/passes.rst:329,337): memory replicated, the clones run on their own copies, the values that
        # leave the region are voted.  The upset sits in ONE copy.
        copies = [[t.clone() for t in inp] for _ in range(nrep)]
        for r in range(runs):
            k = int(rng.integers(0, nrep))
            tgt = flip_memory(eng, copies[k], r, rng)
            tgt["copy"] = k
            targets.append(tgt)
        outs = [bench.run(c, clean) for c in copies]
        width = outs[0].shape[1] * outs[0].element_size()
        w4 = (width + 3) // 4 * 4  # the exit vote works on 32-bit words: every run's row is padded to whole words
        rows8 = [torch.nn.functional.pad(o.contiguous().view(torch.uint8).reshape(runs, width), (0, w4 - width)).contiguous()
                 for o in outs]
        wd = torch.zeros(runs * w4 // 4, dtype=torch.uint8, device="cuda")
        voted = eng.sync_copies(rows8, scrub=True, detected=wd)
        det |= wd.reshape(runs, -1).any(dim=1).to(torch.uint8)
        out = voted[:, :width].contiguous().view(gold.dtype).reshape(runs, -1)
        engine = eng.last_launch()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    st = eng.stats()
    bad = (out.reshape(runs, -1) != gold.reshape(runs, -1)).any(dim=1).cpu().numpy()
    leaked = unexplained = 0
    if a.section == "registers" and bad_staged is not None:
        # wrong matrix m of the staging-only launch: the flip of run m itself, of run m - stride (the workgroup's previous matrix: its last
        # steps stage the next matrix's first slabs, and all of them its f panel), or of run m - 2 * stride (the f piece requested in a
        # matrix's last step is the first piece of the panel after the next: probe of round 4, run 287 -> matrix 415).
        # stride = one matrix per workgroup group = min(runs, CUs / 4).
        # (the kernel's stride between a workgroup's matrices: gridDim.x / 4 = min(batch, CUs / 4))
        stride, staging = min(runs, max(1, cu_count() // 4)), ("s_raw", "f_raw")
        cls_of = lambda r: classes[r] if r >= 0 else None
        for m in np.flatnonzero(bad_staged):
            if cls_of(m - stride) == "f_raw":
                owner = m - stride
            elif cls_of(m - 2 * stride) == "f_raw":
                owner = m - 2 * stride
            elif classes[m] in staging:
                owner = m
            elif cls_of(m - stride) in staging:
                owner = m - stride
            else:
                owner, unexplained = m, unexplained + 1
            leaked += int(owner != m)
            bad[owner] = True
    flagged = det.bool().cpu().numpy()
    hung = np.zeros(runs, dtype=bool)  # quicksort: the watchdog / stack guard cut the run (supervisor: timeout, stack overflow)
    if getattr(bench, "status", None) is not None and a.section == "registers":
        hung = bench.status.cpu().numpy() != 0

    records, counts = [], {"success": 0, "errors": 0, "faults": 0, "timeouts": 0, "invalids": 0, "aborts": 0}
    us = wall * 1e6 / runs
    for r in range(runs):
        if hung[r] and not (aborting and flagged[r]):
            cls, e, f = "timeout", 0, 0
            counts["timeouts"] += 1
        elif aborting and flagged[r]:  # FAULT_DETECTED_DWC() / _CFC() -> abort(): the supervisor logs an abort and a timeout
            cls, e, f = "abort", 0, 0
            counts["timeouts"] += 1
            counts["aborts"] += 1
        elif bad[r]:
            cls, e, f = "error", 1, int(flagged[r])
            counts["errors"] += 1
        elif flagged[r]:
            cls, e, f = "fault", 0, 1
            counts["faults"] += 1
        else:
            cls, e, f = "success", 0, 0
            counts["success"] += 1
        records.append({"run": r, "section": a.section, "target": targets[r],
                        "result": {"core": 0, "errors": e, "faults": f, "runtime_us": us}, "class": cls})
    name = "%s_%s_%s_%s" % (a.benchmark, a.mode, a.section, a.mem_mode if a.section == "memory" else "injector")
    summary = {
        "name": name, "benchmark": a.benchmark, "mode": a.mode, "section": a.section,
        "mem_mode": a.mem_mode if a.section == "memory" else None, "runs": runs,
        "success": counts["success"], "errors": counts["errors"], "faults": counts["faults"],
        "timeouts": counts["timeouts"], "invalids": counts["invalids"], "aborts": counts["aborts"],
        "coverage_pct": 100.0 * (runs - counts["errors"]) / runs,
        "TMR_ERROR_CNT": st["errors_corrected"], "__SYNC_COUNT": st["sync_count"], "dwc_detected": st["dwc_detected"],
        "engine": engine["engine"], "stepwise_blocks": engine["general_blocks"], "hooked_blocks": engine.get("hooked_blocks", 0),
        "wall_s": wall,
        "seconds_per_injection": wall / runs,
        "fault_model": "one single-bit flip of a 32-bit word per run (FaultInjector.flipOneBit, injector.py:202-207)",
        "counters_in_sor": bool(getattr(a, "counters_in_sor", False)),
    }
    if classes is not None:  # the physical register model: outcome per register class, and what the unmodelled share can change
        by = {}
        for r in range(runs):
            d = by.setdefault(classes[r], {"runs": 0, "errors": 0, "corrected_or_masked": 0})
            d["runs"] += 1
            d["errors" if records[r]["class"] == "error" else "corrected_or_masked"] += 1
        unmodelled = by.get("other", {"runs": 0})["runs"]
        summary.update({
            "reg_model": "physical: one bit of one lane of one of the wave's 256 VGPRs, weighted by the register census"
                         + ("; the replica-private classes (acc, b_frag, a_frag) are REAL register flips (COAST_SITE_MM_VGPR), the shared "
                            "staging classes modelled" if a.reg_model == "physical-real" else "")
                         + ("; every modelled class (acc, b_frag, a_frag, s_raw, f_raw) is a REAL register flip (COAST_SITE_MM_VGPR) at a "
                            "random k-slab: a staging register that holds no live word at that moment has no effect" if a.reg_model == "physical-real-all" else ""),
            "census": [{"class": c[0], "vgprs": c[1], "reaches": c[2]} for c in MM.census()],
            "by_class": by, "unmodelled_runs": unmodelled,
            # `other` registers (addresses, lane constants, SGPRs) are not simulated: the two bounds file them under success / error
            "coverage_pct_upper": 100.0 * (runs - counts["errors"]) / runs,
            "coverage_pct_lower": 100.0 * (runs - counts["errors"] - unmodelled) / runs,
        })
        if a.reg_model == "physical-real-all":
            summary["launches"] = "two: every other class's upsets, then the staging flips alone (counters are the sum of both)"
            summary["errors_landed_in_a_later_matrix_of_the_workgroup"] = leaked
            summary["staging_launch_errors_without_a_staging_flip_to_blame"] = unexplained
    else:
        summary["reg_model"] = "sites: replica-private injector sites only (what TMR corrects by construction)" if a.section == "registers" else None
    return records, summary


# ------------------------------------------------------------------------------------------------ the uniform register-file campaign
def uniform_draw(rng, n_vgpr=256, n_sgpr=102, npanels=4, nsteps=16, nwaves=8, nsteps_short=None):
    """one bit of the register state of one wave of the kernel, uniformly: n_vgpr VGPRs x 64 lanes x 32 bits + n_sgpr SGPRs x 32 bits (the
    registers the kernel's code object allocates; the reference's injector draws a register of the core uniformly: injector.py:70-72,
    237-260), at a uniformly random MFMA slot of the panel's 16 steps (60 slots each under TMR, 40 under DWC, 20 unprotected)"""
    vbits, sbits = n_vgpr * 64 * 32, n_sgpr * 32
    file = int(rng.random() < sbits / (vbits + sbits))
    return {"file": file, "reg": int(rng.integers(0, n_sgpr if file else n_vgpr)), "lane": 0 if file else int(rng.integers(0, 64)),
            "bit": int(rng.integers(0, 32)), "wave": (wave := int(rng.integers(0, nwaves))), "panel": int(rng.integers(0, npanels)),
            # (the lane-replica kernel deals its 26 column tiles 7 / 7 / 6 / 6 to four waves: the second half of the waves has fewer steps)
            "step": int(rng.integers(0, nsteps_short if nsteps_short and wave >= nwaves // 2 else nsteps)), "slot": int(rng.integers(0, 60))}


PHYS_KERNEL = "_ZN5coast19mm_mfma_blk3_kernelILi%dELb1ELi2ELb%dEEEvPKjS2_PjjNS_8CountersENS_8FaultTabEPh"  # mm_mfma_blk3_kernel<replicas, true, 2, clone>
PHYS_KERNEL4 = "_ZN5coast19mm_mfma_blk4_kernelILb1ELb%dELi2EEEvPKjS2_PjjNS_8CountersENS_8FaultTabEPh"  # mm_mfma_blk4_kernel<true, clone, 2> (TMR, 128-row panel)
# geometry of the two register-block kernels: rows of a matrix per workgroup, workgroups per matrix, pipeline steps per item
PHYS_KERNEL_LANES = "_ZN5coast20mm_mfma_panel_kernelILi%dELi2EEEvPKjS2_PjjNS_8CountersENS_8FaultTabEPh"  # mm_mfma_panel_kernel<replicas, 2> (replicas in adjacent lanes)
KERNELS = {"blocks3": {"rows": 64, "panels": 4, "steps": 16, "waves": 8, "slots": None},
           "panel128": {"rows": 128, "panels": 2, "steps": 32, "waves": 8, "slots": None},
           # north_star's layout: a workgroup = one 64-row panel of one matrix (not persistent), four waves under TMR (7 / 7 / 6 / 6 column tiles
           # of 8 k-slabs), 20 MFMAs (32 x 32 x 32) per step in every mode
           "lanes": {"rows": 64, "panels": 4, "steps": 56, "steps_short": 48, "waves": 4, "slots": 20}}


def phys_symbol(replicas, clone, kernel="blocks3"):
    if kernel == "lanes":
        return PHYS_KERNEL_LANES % replicas
    if kernel == "panel128":
        if replicas != 3:
            raise SystemExit("campaign: mm_mfma_blk4_kernel (--kernel panel128) is the TMR kernel; DWC / unprotected run mm_mfma_blk3_kernel")
        return PHYS_KERNEL4 % int(bool(clone))
    return PHYS_KERNEL % (replicas, int(bool(clone)))


def _kernel_text(sym):
    """(disassembly, metadata notes) of one kernel of the library that runs: the gfx950 code object out of the bundle in its .hip_fatbin,
    through the image's llvm-objdump / llvm-readelf"""
    import struct
    import subprocess
    import tempfile

    from coast_amd import _lib as libmod

    path = os.environ.get("COAST_LIB_OVERRIDE") or libmod.lib_path()
    data = open(path, "rb").read()
    # (the library is several translation units, each with an offload bundle of its own in .hip_fatbin: look for the kernel in every one)
    cos, pos = [], data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    if pos < 0:
        raise SystemExit("campaign: no offload bundle in %s" % path)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        off = pos + 32
        for _ in range(n):
            o, sz, ts = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + ts].decode()
            off += 24 + ts
            if "gfx950" in triple:
                cos.append(data[pos + o:pos + o + sz])
        pos = data.find(b"__CLANG_OFFLOAD_BUNDLE__", pos + 1)
    if not cos:
        raise SystemExit("campaign: no gfx950 code object in %s" % path)
    llvm = "/opt/rocm/lib/llvm/bin/"
    text, notes = "", ""
    for co in cos:
        if sym.encode() not in co:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as fh:
            fh.write(co)
            fh.flush()
            text = subprocess.run([llvm + "llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + sym, fh.name],
                                  capture_output=True, text=True, check=True).stdout
            notes = subprocess.run([llvm + "llvm-readelf", "--notes", fh.name], capture_output=True, text=True, check=True).stdout
        if "v_mfma" in text:
            break
    if "v_mfma" not in text:
        raise SystemExit("campaign: %s not found in the code object" % sym)
    return text, notes


def kernel_registers(replicas, clone=False, kernel="blocks3"):
    """(VGPRs, SGPRs the kernel allocates, the VGPRs it spills scalar registers into) of mm_mfma_blk3_kernel<replicas, true, 2, clone>, read
    from the code object inside the very library that runs (the bundle in its .hip_fatbin) with the image's llvm-readelf / llvm-objdump: no
    allocation map to keep in step with the compiler.  The spill registers (v_writelane_b32 / v_readlane_b32): a lane of such a register IS a
    scalar register -- a descriptor word, a kernel-argument pointer, a loop counter -- and an upset there belongs to the scalar class (it can
    send an access anywhere in the address space)."""
    import re

    sym = phys_symbol(replicas, clone, kernel)
    text, notes = _kernel_text(sym)
    rec = [blk for blk in notes.split("  - .agpr_count:") if ".name:           " + sym + "\n" in blk]
    if len(rec) != 1:
        raise SystemExit("campaign: no metadata record of %s" % sym)
    nv, ns = int(re.search(r"\.vgpr_count:\s+(\d+)", rec[0]).group(1)), int(re.search(r"\.sgpr_count:\s+(\d+)", rec[0]).group(1))
    return nv, min(ns, 102), sorted({int(m) for m in re.findall(r"v_writelane_b32\s+v(\d+)", text)})


def kernel_addend_tuples(replicas, slot, clone=False, kernel="blocks3"):
    """for every step body of mm_mfma_blk3_kernel<replicas, true, 2, clone> (a body = 20 x replicas consecutive MFMAs of the disassembly): the
    register tuple (first, last) the MFMA of slot `slot` takes as its addend -- a limb-sum accumulator that is LIVE in front of that slot,
    whatever the allocator does with it afterwards (the accumulators migrate: a v_mfma's destination need not be its addend).  Read from the
    running library's own code object.  (tests: the PREG decode)"""
    import re

    text, _ = _kernel_text(phys_symbol(replicas, clone, kernel))
    per = 20 if kernel == "lanes" else 20 * replicas  # (the lane-replica kernel: 20 MFMAs of 32 x 32 x 32 per step in every mode, 16-register tuples)
    op = "v_mfma_i32_32x32x32_i8" if kernel == "lanes" else "v_mfma_i32_16x16x64_i8"
    mf = re.findall(op + r"\s+v\[\d+:\d+\],\s*v\[\d+:\d+\],\s*v\[\d+:\d+\],\s*(?:v\[(\d+):(\d+)\]|\S+)", text)
    out = []
    for b in range(len(mf) // per):
        c0, c1 = mf[b * per + slot]
        if c0:
            out.append((int(c0), int(c1)))
    return out


def preg_row(item, d):
    # (steps 16..31 of mm_mfma_blk4_kernel's 32-step items: the fifth bit of the step rides in bit 29)
    step = d["slot"] | ((d["step"] & 15) << 6) | (d["lane"] << 10) | (d["wave"] << 16) | (d["file"] << 19) | (d["reg"] << 20) | (((d["step"] >> 4) & 3) << 29)
    return (item, 0, ca.SITE_MM_PREG, step, d["bit"])


def _no_core_files():
    import resource

    resource.setrlimit(resource.RLIMIT_CORE, (0, 0))


def run_uniform_campaign(a, eng=None):
    """--reg-model uniform (-b mm --side 256): ONE coverage figure for the matrix-core kernel, no census, no model.  A run = one workgroup
    group (four 64-row panels) that multiplies three matrices in a row; the upset -- COAST_SITE_MM_PREG: an exclusive-or on one bit of a
    physical register, v0..v255 of one lane or s0..s101, of one wave of one of the four workgroups -- falls in front of a uniformly random
    MFMA slot of the MIDDLE matrix; the run is an error when ANY word of the group's three products is wrong (a staged word, an address
    register or a loop counter can carry the damage into the next matrix; nothing carries it into another workgroup).  64 runs per launch
    (one per workgroup group of a 256-CU part), so a run's outcome is its own.
    The launches run in CHILD processes (as every run of the reference's campaign is a QEMU process of its own): an upset can turn a pointer
    into a wild address -- a scalar register, or a lane of the vector registers the compiler spills scalar registers into -- and the memory
    fault that follows takes the process with it.  A launch whose child dies is split until the run that kills it is alone: that run is
    `invalid` (supervisor.py files a crashed run the same way) and counts against the coverage.  Scalar-class upsets (s0..s101 and the spill
    registers' lanes): --sgpr count (default) does not execute them and counts every one as an error; --sgpr run executes them, one per launch."""
    import subprocess

    rng = np.random.default_rng(a.seed)
    rep = MODES[a.mode]
    if a.benchmark != "mm" or a.side != 256 or a.mode == "CFCSS":
        raise SystemExit("--reg-model uniform: the register file is that of the matrix-core kernel (-b mm --side 256 -m TMR | DWC | NONE)")
    # (DWC and the unprotected mode run mm_mfma_blk3_kernel whatever the TMR register-block kernel is; the lane-replica kernel serves all three)
    kern = a.kernel if (rep == ca.TMR or a.kernel == "lanes") else "blocks3"
    if kern == "lanes" and rep != ca.TMR:
        raise SystemExit("--kernel lanes: -m TMR (the DWC / unprotected wave counts and tile deals of mm_mfma_panel_kernel are not wired into the draw)")
    geo = KERNELS[kern]
    quads = max(1, cu_count() // geo["panels"])  # workgroup groups of a launch = its stride between a workgroup's matrices
    runs = a.runs
    nv, ns, spill = kernel_registers(max(rep, 1), a.clone_staging and rep > 1, kern)
    spill = set(spill)  # vector registers whose lanes hold spilled scalar registers: the scalar class too
    nslots = geo["slots"] or 20 * max(rep, 1)  # MFMA slots of a pipeline step
    draws = [uniform_draw(rng, nv, ns, geo["panels"], geo["steps"], geo["waves"], geo.get("steps_short")) for _ in range(runs)]
    for d in draws:
        d["slot"] %= nslots
    scalar = lambda d: d["file"] == 1 or d["reg"] in spill
    executed = np.array([not scalar(d) or a.sgpr == "run" for d in draws])
    vec = [r for r in range(runs) if not scalar(draws[r])]
    if os.environ.get("COAST_CAMPAIGN_ONLY"):  # development: arm the runs lo <= r < hi only
        lo, hi = (int(x) for x in os.environ["COAST_CAMPAIGN_ONLY"].split(":"))
        vec = [r for r in vec if lo <= r < hi]
    todo = [vec[k:k + quads] for k in range(0, len(vec), quads)]
    if a.sgpr == "run":
        todo += [[r] for r in range(runs) if scalar(draws[r])]
    results, invalid, stats = {}, set(), {"errors_corrected": 0, "sync_count": 0, "dwc_detected": 0}
    t0 = time.perf_counter()
    children = 0
    while todo:
        spec = json.dumps({"mode": a.mode, "seed": a.seed, "clone": bool(a.clone_staging), "kernel": kern, "launches": [[[r, draws[r]] for r in grp] for grp in todo]})
        children += 1
        # (a child that an upset takes down must not leave a core file behind: with 12 GB of device memory mapped each one is tens of GB, and
        # the scalar-class runs of --sgpr run filled a 79 GB disk in one campaign)
        dbg = os.environ.get("COAST_CAMPAIGN_DEBUG", "")  # development: "t" = per-child wall time and exit status on stderr; "e" = children keep the parent's environment
        tc = time.perf_counter()
        proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--preg-child", "-"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                                preexec_fn=_no_core_files, env=dict(os.environ) if "e" in dbg else dict(os.environ, HSA_ENABLE_COREDUMP="0"))
        try:
            # (a launch takes a tenth of a second; the first child of a fresh box pages torch in for a minute or two.  A child that hangs --
            # an upset that turns a loop bound into a long walk -- is cut and its launch halved like a crashed one's)
            # (--child-timeout T: a caller whose own process has torch and the device warm -- the test suite -- cuts a hung child after
            # T + one second per launch; a campaign's three or four hangs otherwise cost it 250 + 3 x 86 s)
            base = a.child_timeout if a.child_timeout else (240 if children == 1 else 75)
            out, _ = proc.communicate(spec, timeout=base + len(todo))
        except subprocess.TimeoutExpired:
            proc.kill()
            out, _ = proc.communicate()
        if "t" in dbg:
            print("child %d: %d launches, %.1f s, exit %s" % (children, len(todo), time.perf_counter() - tc, proc.returncode), file=sys.stderr, flush=True)
        started, done = None, set()
        for line in out.splitlines():
            w = line.split(" ", 2)
            if w[0] == "start":
                started = int(w[1])
            elif w[0] == "done":
                k = int(w[1])
                res = json.loads(w[2])
                for r, bd, fl in res["runs"]:
                    results[r] = (bool(bd), bool(fl))
                for key in stats:
                    stats[key] += res["stats"][key]
                done.add(k)
                started = None
        rest = [grp for k, grp in enumerate(todo) if k not in done]
        if started is not None and started not in done:  # the child died inside launch `started`: halve it until the killer is alone
            grp = todo[started]
            rest = [g for g in rest if g is not grp]
            if len(grp) == 1:
                invalid.add(grp[0])
            else:
                rest = [grp[:len(grp) // 2], grp[len(grp) // 2:]] + rest
        elif proc.returncode != 0 and len(rest) == len(todo):
            raise SystemExit("campaign: the child made no progress (exit %s):\n%s" % (proc.returncode, out[-2000:]))
        todo = rest
    wall = time.perf_counter() - t0
    aborting = rep == ca.DWC
    records, counts = [], {"success": 0, "errors": 0, "faults": 0, "timeouts": 0, "invalids": 0, "aborts": 0, "not_executed": 0}
    by_reg = {}
    for r, d in enumerate(draws):
        bad, flagged = results.get(r, (False, False))
        if r in invalid:
            cls = "invalid"
            counts["invalids"] += 1
        elif not executed[r] or r not in results:
            cls = "error"  # not executed, counted against the coverage
            counts["errors"] += 1
            counts["not_executed"] += 1
        elif aborting and flagged:
            cls = "abort"
            counts["timeouts"] += 1
            counts["aborts"] += 1
        elif bad:
            cls = "error"
            counts["errors"] += 1
        elif flagged:
            cls = "fault"
            counts["faults"] += 1
        else:
            cls = "success"
            counts["success"] += 1
        key = ("s%d" if d["file"] else "v%d") % d["reg"]
        e = by_reg.setdefault(key, [0, 0])
        e[0] += 1
        e[1] += cls in ("error", "invalid")
        records.append({"run": r, "section": "registers", "target": d, "class": cls,
                        "result": {"core": 0, "errors": int(cls == "error"), "faults": int(bool(flagged)), "runtime_us": wall * 1e6 / runs}})
    nbad = counts["errors"] + counts["invalids"]
    summary = {
        "name": "mm_%s_registers_uniform%s%s" % (a.mode, "_clone_staging" if a.clone_staging else "", "_" + kern if kern != "blocks3" else ""), "clone_staging": bool(a.clone_staging),
        "kernel": {"panel128": "mm_mfma_blk4_kernel (128-row panel)", "blocks3": "mm_mfma_blk3_kernel (64-row panel)", "lanes": "mm_mfma_panel_kernel (replicas in adjacent lanes)"}[kern], "benchmark": "mm", "mode": a.mode, "section": "registers", "mem_mode": None, "runs": runs,
        "success": counts["success"], "errors": counts["errors"], "faults": counts["faults"], "timeouts": counts["timeouts"],
        "invalids": counts["invalids"], "aborts": counts["aborts"], "scalar_upsets_not_executed_counted_as_errors": counts["not_executed"],
        "coverage_pct": 100.0 * (runs - nbad) / runs,
        # (an upset of a tally register -- agree, the vote counts -- lands in these sums as whatever the flipped bit weighs)
        "TMR_ERROR_CNT": stats["errors_corrected"], "__SYNC_COUNT": stats["sync_count"], "dwc_detected": stats["dwc_detected"],
        "engine": "matrix_core", "stepwise_blocks": 0, "hooked_blocks": 0, "child_processes": children,
        "wall_s": wall, "seconds_per_injection": wall / runs,
        "fault_model": "one single-bit flip of a 32-bit register per run (FaultInjector.flipOneBit, injector.py:202-207); the register is drawn uniformly "
                       "from the wave's register state (injector.py:70-72, 237-260): v0..v%d x 64 lanes, s0..s%d (what the kernel's code object allocates)" % (nv - 1, ns - 1),
        "reg_model": "uniform: COAST_SITE_MM_PREG, a real exclusive-or on a PHYSICAL register of the running kernel in front of a uniformly random MFMA slot "
                     "of the middle one of the three matrices a workgroup group multiplies; error = any wrong word in the three products; invalid = "
                     "the process died (memory fault)",
        "registers_with_errors": {k: v for k, v in sorted(by_reg.items(), key=lambda kv: -kv[1][1]) if v[1]},
        "scalar_class": "s0..s101 and the lanes of %s (the vector registers the compiler spills scalar registers into: v_writelane_b32 in the "
                        "kernel's code object)" % ", ".join("v%d" % r for r in sorted(spill)),
        "counters_in_sor": False,
    }
    return records, summary


def preg_child(_):
    """one child of the uniform campaign: the launches of the spec on stdin, `start k` before and `done k {...}` behind each"""
    spec = json.loads(sys.stdin.read())
    eng = ca.Engine(0)
    n, nn, items = 256, 256 * 256, 3
    kern = spec.get("kernel", "blocks3")
    geo = KERNELS[kern]
    os.environ["COAST_MM_TILE"] = kern  # (read by the library at every call)
    quads = max(1, cu_count() // geo["panels"])
    g = torch.Generator(device="cuda").manual_seed(spec["seed"])
    f, s = [torch.randint(-2**31, 2**31, (items * quads, n, n), dtype=torch.int32, device="cuda", generator=g) for _ in range(2)]
    gold = eng.mm_batch(f, s, cfg=ca.XmrConfig(ca.UNPROTECTED)).clone()
    cfg = ca.XmrConfig(MODES[spec["mode"]], 0, 0 if spec.get("clone") else ca.F_SINGLE_STAGING)
    for k, grp in enumerate(spec["launches"]):
        print("start %d" % k, flush=True)
        eng.reset_stats()
        eng.inject_faults(ca.make_faults([preg_row((quads + q) * nn + geo["rows"] * d["panel"] * n, d) for q, (r, d) in enumerate(grp)]))
        det = torch.zeros(items * quads * nn, dtype=torch.uint8, device="cuda")
        out = eng.mm_batch(f, s, cfg=cfg, detected=det)
        wrong = (out != gold).reshape(items, quads, -1).any(dim=2).any(dim=0).cpu().numpy()
        seen = det.reshape(items, quads, -1).any(dim=2).any(dim=0).cpu().numpy()
        st = eng.stats()
        print("done %d %s" % (k, json.dumps({"runs": [[r, int(wrong[q]), int(seen[q])] for q, (r, d) in enumerate(grp)],
                                             "stats": {key: st[key] for key in ("errors_corrected", "sync_count", "dwc_detected")}})), flush=True)


def format_summary(s):
    """FileSummary.__str__ (jsonParser.py:46-75)"""
    n = s["runs"]
    good = s["success"] + s["faults"]
    lines = ["", ("Summary for file %s:" % s["name"]).center(60), "",
             "Total runs: %d" % n,
             "Successes:  %d (%3.2f%%)" % (good, 100.0 * good / n),
             "Errors:     %d (%3.2f%%)" % (s["errors"], 100.0 * s["errors"] / n),
             "Faults:     %d (%3.2f%%)" % (s["faults"], 100.0 * s["faults"] / n),
             "Timeouts:   %d (%3.2f%%)" % (s["timeouts"], 100.0 * s["timeouts"] / n),
             "Invalid:    %d (%3.2f%%)" % (s["invalids"], 100.0 * s["invalids"] / n),
             "Time to run: %s" % str(datetime.timedelta(seconds=int(s["wall_s"]))),
             " (%3.6f seconds per injection)" % s["seconds_per_injection"]]
    if s["aborts"]:
        lines += ["Additional Data:", "Aborts:     %d (%3.2f%%)" % (s["aborts"], 100.0 * s["aborts"] / n)]
    if "registers_with_errors" in s:
        top = list(s["registers_with_errors"].items())[:24]
        lines += ["Scalar upsets not executed (counted as errors): %d" % s["scalar_upsets_not_executed_counted_as_errors"],
                  "Coverage:   %3.2f%% (uniform draw over the wave's register state; one figure)" % s["coverage_pct"],
                  "Registers with errors (runs, errors): " + ", ".join("%s %d/%d" % (k, v[1], v[0]) for k, v in top)]
    if s.get("by_class"):
        lines += ["Register classes (physical model):"]
        for c in s["census"]:
            d = s["by_class"].get(c["class"], {"runs": 0, "errors": 0})
            lines.append("  %-8s %3d VGPRs  runs %5d  errors %5d" % (c["class"], c["vgprs"], d["runs"], d["errors"]))
        lines.append("Coverage:   %3.2f%% .. %3.2f%% (unmodelled registers counted as errors .. as successes)"
                     % (s["coverage_pct_lower"], s["coverage_pct_upper"]))
    return "\n".join(lines)


def write_logs(a, records, summary):
    os.makedirs(a.log_dir, exist_ok=True)
    now = datetime.datetime.now()
    ts = "%d-%d-%02d_%02d-%02d" % (now.year, now.month, now.day, now.hour, now.minute)
    prefix = os.path.join(a.log_dir, "mi355x_%s_%s" % (summary["name"], ts))
    with open(prefix + ".log", "w") as fh:  # the UART line of every run (decoder.py:66)
        for rec in records:
            res = rec["result"]
            fh.write("run %d  %s  C:%d E:%d F:%d T:%dus%s\n" % (rec["run"], json.dumps(rec["target"], sort_keys=True), res["core"],
                                                                res["errors"], res["faults"], int(res["runtime_us"]),
                                                                "  ABORT (FAULT_DETECTED)" if rec["class"] == "abort" else ""))
        fh.write(format_summary(summary) + "\n")
    with open(prefix + ".json", "w", encoding="utf-8") as fh:
        json.dump({"summary": summary, "runs": records}, fh)
    return prefix


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("-b", "--benchmark", default="mm", choices=sorted(BENCHES))
    ap.add_argument("-m", "--mode", default="TMR", choices=list(MODES))
    ap.add_argument("-t", "--runs", type=int, default=5000)
    ap.add_argument("-s", "--section", default="registers", choices=["registers", "memory"])
    ap.add_argument("--mem-mode", default="nomemrep", choices=["nomemrep", "default", "storesync"])
    ap.add_argument("--reg-model", default="sites", choices=["sites", "physical", "physical-real", "physical-real-all", "uniform"],
                    help="registers: `sites` = a replica-private injector site per run; `physical` (-b mm --side 256 -m TMR) = any "
                         "register of the matrix-core kernel's wave, weighted by its register census, shared state included; "
                         "`physical-real` = the replica-private classes as real flips of the running kernel's VGPRs; "
                         "`physical-real-all` = the staging registers as real flips too (an error can land in the workgroup's next matrix); "
                         "`uniform` = one bit of ANY physical register of a wave (COAST_SITE_MM_PREG), drawn uniformly: one coverage figure")
    ap.add_argument("--sgpr", default="count", choices=["count", "run"],
                    help="--reg-model uniform: scalar-register upsets are not executed and count as errors (count), or run one per launch in "
                         "child processes (run: a wild descriptor is a memory fault that ends the child)")
    ap.add_argument("--clone-staging", action="store_true",
                    help="-b mm --side 256: run with the staging loads cloned and compared (the library's default since ABI 8; without this option the campaign passes COAST_F_SINGLE_STAGING)")
    ap.add_argument("--kernel", default="panel128", choices=["panel128", "blocks3", "lanes"],
                    help="--reg-model uniform -m TMR: the matrix-core kernel whose register file is drawn from -- mm_mfma_blk4_kernel (128-row panel, "
                         "the library's TMR default since round 6) or mm_mfma_blk3_kernel (64-row panel, rounds 4-5; always the DWC / unprotected kernel)")
    ap.add_argument("--preg-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--child-timeout", type=int, default=0,
                    help="--reg-model uniform: seconds (+ one per launch) before a child process is taken for hung and its launch halved "
                         "(default: 240 for the first child of a run -- a fresh box pages torch in --, 75 for the others)")
    ap.add_argument("--counters-in-sor", action="store_true",
                    help="registers: run with COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC (the loop counters replica-private, their branch conditions "
                         "and GEP offsets voted) and aim every upset at a loop counter")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--side", type=int, default=9, help="mm: matrix side, one matrix per run (256 = the matrix-core engine)")
    ap.add_argument("--chaes-type", type=int, default=128128, help="chaes: key bits * 1000 + block bits (aes_key.c:83-134)")
    ap.add_argument("-l", "--log-dir", default="./logs/")
    ap.add_argument("-n", "--no-logging", action="store_true")
    return ap.parse_args(argv)


def main():
    a = parse()
    if a.preg_child:
        return preg_child(a.preg_child)
    records, summary = run_uniform_campaign(a) if a.reg_model == "uniform" else run_campaign(a)
    if not a.no_logging:
        summary["log_prefix"] = write_logs(a, records, summary)
    print(format_summary(summary))
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
