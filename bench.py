#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: protected elems/sec + corrected-fault count,
matrixMultiply TMR (configs[1]: 256x256 uint32, 3-lane replicate + vote), 1..8 GPUs of one node.

A step = one pass of the protected hot path over one batch of synthetic inputs that are already resident in HBM:
arm the on-device injector with a seeded fault list, run the protected kernel on this GPU's shard of independent
blocks, fold the fault counters and all-reduce them across GPUs (RCCL; the only collective of the path).  Blocks shard
across ranks with no data exchange, so scaling is weak (per-GPU batch fixed).

Default workload = mm (the metric BASELINE.json quotes).  `--workload crc16|sha256|aes` runs the other BASELINE configs
through the same harness (development / north-star targets; the driver only runs the default).

Prints ONE JSON line (rank 0).  The oracle is used only for the cpu_baseline leg.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# Integer-MAC issue ceiling of the chip.  MI355X_MICROARCH.md gives the full-rate VALU figure (256 CU x 4 SIMD x 32 lanes
# x 2.4 GHz = 78.6 T lane-ops/s = 157.3 TFLOP/s / 2) but no integer-multiply rates, so the MAC peak is the measured
# issue rate of v_mad_u64_u32 -- the one-instruction 32-bit MAC the kernel is built from -- at 8 waves/SIMD:
# 5.11 cycles per wave-instruction = 30.78 T lane-MAC/s (tools/valu_microbench, profiles/microbench_r01.txt).
MAC_PEAK = 30.78e12
# int8 MFMA: MI355X_MICROARCH.md lists no spec figure, "~2x bf16 rate" (bf16 ~2.5 PFLOP/s dense) and a ubench ceiling of
# >= 3944 TOPS; tools/mfma_probe reaches 4.2-4.3 POPS with v_mfma_i32_32x32x32_i8 (32 cycles/instruction/SIMD at the
# ~2.05 GHz the chip sustains under MFMA load).  Peak = 2 x 2.5e15; the measured ceiling is reported next to it.
I8_MFMA_PEAK = 5.0e15
I8_MFMA_UBENCH = 4.25e15
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9  # 78.6 T lane-ops/s, full-rate VALU


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="mm", choices=["mm", "crc16", "sha256", "aes", "cache_test", "chsha"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU items per step (0 = the BASELINE config's size)")
    ap.add_argument("--side", type=int, default=256)
    ap.add_argument("--faults", type=int, default=-1,
                    help="single-bit flips injected per GPU per step (default: 4096 for mm, 1024 otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ CPU baselines
def _time_budget(fn, budget_s):
    t0 = time.perf_counter()
    reps = 0
    while True:
        fn()
        reps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return reps, time.perf_counter() - t0


def cpu_baseline_mm(side, budget_s=12.0):
    """Default-mode CPU-TMR restatement of matrix_multiply (oracle/cpu_tmr_baseline.c), single thread -- the
    reference is single-threaded by construction.  Bounded sample: as many side x side matrices as fit the budget."""
    from oracle import oracle as orc

    orc.build()
    rng = np.random.default_rng(0)
    f = rng.integers(0, 2**32, (side, side), dtype=np.uint32)
    s = rng.integers(0, 2**32, (side, side), dtype=np.uint32)
    gold = orc.mm_xor(orc.mm_plain(f, s))

    def one():
        r, err, cnt, syncs = orc.cpu_tmr_mm(f, s, gold)
        assert err == 0 and cnt == 0

    reps, dt = _time_budget(one, budget_s)
    ureps, du = _time_budget(lambda: orc.mm_plain(f, s), 2.0)  # unprotected arithmetic, for the CPU TMR overhead
    # all host cores over independent matrices (the reference itself is single-threaded; this is the generous bound).
    # Visible CPUs can exceed what the container may use, so calibrate with one matrix per thread and size the timed
    # run for ~4 s of wall time.
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    w1 = orc.cpu_tmr_mm_threads(f, s, gold, ncpu, 1)
    per_thread = max(1, min(64, int(4.0 / max(w1, 1e-3))))
    wall = orc.cpu_tmr_mm_threads(f, s, gold, ncpu, per_thread)
    return {
        "value": reps * side * side / dt, "unit": "protected elems/s", "cores": 1, "kind": "port",
        "sample": "%d x (%dx%d uint32 matrix_multiply + checkGolden), default-mode TMR restatement "
                  "(memory x3, loop-condition votes, -countErrors), gcc -O3, %.1f s" % (reps, side, side, dt),
        "unprotected_elems_per_s": ureps * side * side / du,
        "tmr_overhead_x": (dt / reps) / (du / ureps),
        "host_cpus": ncpu,
        "all_cores": {"value": ncpu * per_thread * side * side / wall if wall > 0 else None, "cores": ncpu,
                      "unit": "protected elems/s",
                      "sample": "%d threads x %d matrices, %.1f s wall" % (ncpu, per_thread, wall)},
    }


def cpu_baseline_items(kind, budget_s=10.0):
    """The oracle's replicated model (-noMemReplication schedule, the one the GPU instantiates) timed on one core."""
    from oracle import oracle as orc

    orc.build()
    rng = np.random.default_rng(0)
    if kind == "crc16":
        data = rng.integers(0, 256, (1 << 14, 256), dtype=np.uint8)
        reps, dt = _time_budget(lambda: orc.crc16_xmr(data, 256, replicas=3), budget_s)
        return {"value": reps * data.size / dt * 1e-9, "unit": "GB/s", "cores": 1, "kind": "port",
                "sample": "%d x 4 MiB (16384 blocks x 256 B), oracle TMR model, gcc -O3, %.1f s" % (reps, dt)}
    if kind == "sha256":
        msgs = rng.integers(0, 256, (1 << 15, 64), dtype=np.uint8)
        reps, dt = _time_budget(lambda: orc.sha256_xmr(msgs, 64, replicas=3), budget_s)
        return {"value": reps * msgs.shape[0] / dt, "unit": "msgs/s", "cores": 1, "kind": "port",
                "sample": "%d x 32768 messages x 64 B, oracle TMR model, gcc -O3, %.1f s" % (reps, dt)}
    if kind == "chsha":
        msgs = rng.integers(0, 256, (1 << 9, 16384), dtype=np.uint8)
        reps, dt = _time_budget(lambda: orc.chsha_xmr(msgs, 16384, replicas=3), budget_s)
        return {"value": reps * msgs.size / dt * 1e-9, "unit": "GB/s", "cores": 1, "kind": "port",
                "sample": "%d x 512 messages x 16 KiB, oracle TMR model, gcc -O3, %.1f s" % (reps, dt)}
    if kind == "cache_test":
        arrs = np.tile(np.arange(600, dtype=np.int32), (1 << 13, 1))
        reps, dt = _time_budget(lambda: orc.cache_test_xmr(arrs, replicas=3), budget_s)
        return {"value": reps * arrs.size * 4 / dt * 1e-9, "unit": "GB/s", "cores": 1, "kind": "port",
                "sample": "%d x 8192 arrays x 600 ints, oracle TMR model, gcc -O3, %.1f s" % (reps, dt)}
    st = rng.integers(0, 256, (1 << 15, 16), dtype=np.uint8)
    key = rng.integers(0, 256, (1 << 15, 16), dtype=np.uint8)
    reps, dt = _time_budget(lambda: orc.aes128_xmr(st, key, 0, replicas=2), budget_s)
    return {"value": reps * st.shape[0] / dt, "unit": "blocks/s", "cores": 1, "kind": "port",
            "sample": "%d x 32768 blocks encrypt, oracle DWC model, gcc -O3, %.1f s" % (reps, dt)}


# ------------------------------------------------------------------------------------------------ workloads
class Workload:
    """setup() allocates device-resident synthetic inputs; launch() enqueues one protected pass."""


class MM(Workload):
    metric = "protected elems/sec + corrected-fault count, matrixMultiply TMR"
    unit = "protected elems/s"
    dtype = "u32"

    def __init__(self, a, eng, dev, rank, coast_amd):
        self.n, self.batch = a.side, a.batch or 16384  # 12.9 GB of f, s, r: ~32 ms kernels (a batch, not one matrix)
        n, batch = self.n, self.batch
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        self.f = torch.randint(-2**31, 2**31, (batch, n, n), dtype=torch.int32, device=dev, generator=g)
        self.s = torch.randint(-2**31, 2**31, (batch, n, n), dtype=torch.int32, device=dev, generator=g)
        self.r = torch.empty_like(self.f)
        self.cfg = coast_amd.XmrConfig(coast_amd.TMR)
        self.eng, self.ca = eng, coast_amd
        # one accumulator upset in one replica of K distinct output elements: each must be out-voted and counted once
        rng = np.random.default_rng(99 + rank)
        items = rng.choice(batch * n * n, a.faults, replace=False)
        self.faults = coast_amd.make_faults([(int(it), int(rng.integers(0, 3)), coast_amd.SITE_MM_ACC,
                                              int(rng.integers(0, n + 1)), int(rng.integers(0, 32))) for it in items])
        self.units_per_step = batch * n * n

    def launch(self):
        self.eng.mm_batch(self.f, self.s, out=self.r, cfg=self.cfg)

    def check(self):
        chk = self.eng.mm_batch(self.f[:2].contiguous(), self.s[:2].contiguous(),
                                cfg=self.ca.XmrConfig(self.ca.UNPROTECTED))
        return bool(torch.equal(chk, self.r[:2]))

    def config(self, world):
        return {"workload": "matrixMultiply %dx%d uint32 TMR (3-lane replicate + vote), batch %d matrices/GPU, "
                            "%d injected single-bit faults/GPU/step" % (self.n, self.n, self.batch, len(self.faults)),
                "side": self.n, "batch_per_gpu": self.batch, "replicas": 3, "engine": self.engine(),
                "parallelism": "dp%d (independent matrices)" % world}

    def engine(self):
        """which kernel coast_mm_batch dispatches to (coast_hip.hip LAUNCH_MM): side 256 runs on the matrix cores"""
        return "mfma" if self.n == 256 and os.environ.get("COAST_MM_ENGINE") != "valu" else "valu"

    def roofline(self, kern_ms):
        n, batch = self.n, self.batch
        macs = float(batch) * n ** 3           # algorithmic 32-bit MACs of one launch (N^3 per matrix)
        bytes_alg = float(batch) * 12 * n * n  # algorithmic HBM bytes of one launch (read f, s once; write r once)
        t = kern_ms * 1e-3
        hbm = {"hbm_achieved_GBs": bytes_alg / t * 1e-9, "hbm_peak_GBs": HBM_PEAK_GBS,
               "hbm_frac": bytes_alg / t * 1e-9 / HBM_PEAK_GBS, "algorithmic_bytes": bytes_alg, "kernel_ms": kern_ms}
        if self.engine() == "mfma":
            # a wrapping 32-bit MAC = 10 signed-byte limb products (p+q <= 3), per replica: the int8 work the protected
            # computation needs on the matrix core (lane padding 32/30 and ragged tiles are NOT counted)
            ops = 2.0 * macs * 10 * 3
            return dict(hbm, **{
                "bound": "mfma", "kernel": "mm_mfma_panel_kernel<3>",
                "achieved": ops / t * 1e-12, "peak": I8_MFMA_PEAK * 1e-12, "unit": "TOP/s (int8)",
                "frac": ops / t / I8_MFMA_PEAK, "frac_of_ubench_ceiling": ops / t / I8_MFMA_UBENCH,
                "u32_macs_per_s": macs / t, "int8_ops_per_u32_mac": 2 * 10 * 3,
                # SURVEY 8(d) priced this path against the VALU MAC ceiling (N^3 algorithmic MACs per matrix, x3 executed):
                "algorithmic_macs_vs_valu_ceiling": macs / t / MAC_PEAK, "executed_macs_vs_valu_ceiling": 3.0 * macs / t / MAC_PEAK,
                "note": "r = sum_k f*s mod 2^32 as ten int8 GEMMs of signed-byte limbs on v_mfma_i32_32x32x32_i8, x3 replicas in "
                        "adjacent lane-columns; achieved = 2*N^3*10*3 int8 ops per matrix / kernel time; peak = 2x the bf16 dense "
                        "peak (MI355X_MICROARCH.md: I8 runs at ~2x bf16 rate; ubench ceilings 3944 there, 4.2-4.3 POPS in "
                        "tools/mfma_probe at the ~2.05 GHz the chip holds under MFMA load)",
            })
        return dict(hbm, **{
            "bound": "valu", "kernel": "mm_fast256_kernel<3>" if n == 256 else "mm_fast_kernel<3>",
            "achieved": macs / t * 1e-12, "peak": MAC_PEAK * 1e-12, "unit": "T int32-MAC/s",
            "frac": macs / t / MAC_PEAK, "executed_frac": 3.0 * macs / t / MAC_PEAK,
            "note": "VALU engine: bound = issue of v_mad_u64_u32 (measured 30.78 T lane-MAC/s, profiles/microbench_r01.txt); "
                    "achieved/frac count ALGORITHMIC MACs (N^3 per matrix), TMR executes 3x of them (executed_frac)",
        })

    def cpu(self):
        return cpu_baseline_mm(self.n)


class CRC16(Workload):
    metric = "protected bytes/sec + corrected-fault count, crc16 TMR stream"
    unit = "GB/s"
    dtype = "u16"

    def __init__(self, a, eng, dev, rank, coast_amd):
        self.bl = 256
        self.nb = a.batch or (1 << 25)  # 2^25 blocks x 256 B = 8 GiB per GPU (64 GiB over 8 GPUs)
        g = torch.Generator(device=dev).manual_seed(16 + rank)
        self.data = torch.empty(self.nb * self.bl, dtype=torch.uint8, device=dev)
        step = 1 << 28
        for off in range(0, self.data.numel(), step):  # generated on-device, shard by shard
            self.data[off:off + step].copy_(torch.randint(0, 256, (min(step, self.data.numel() - off),),
                                                          dtype=torch.uint8, device=dev, generator=g))
        self.out = torch.empty(self.nb, dtype=torch.int16, device=dev)
        self.cfg = coast_amd.XmrConfig(coast_amd.TMR)
        self.eng = eng
        rng = np.random.default_rng(7 + rank)
        items = rng.choice(self.nb, a.faults, replace=False)
        self.faults = coast_amd.make_faults([(int(it), int(rng.integers(0, 3)), coast_amd.SITE_CRC_CRC,
                                              int(rng.integers(0, self.bl + 1)), int(rng.integers(0, 16))) for it in items])
        self.units_per_step = self.nb * self.bl * 1e-9  # GB

    def launch(self):
        self.eng.crc16_batch(self.data, self.bl, out=self.out, cfg=self.cfg)

    def check(self):
        import coast_amd

        ref = self.eng.crc16_batch(self.data[: 4096 * self.bl], self.bl, cfg=coast_amd.XmrConfig(coast_amd.UNPROTECTED))
        return bool(torch.equal(ref, self.out[:4096]))

    def config(self, world):
        return {"workload": "crc16 %d-byte blocks TMR, %.1f GiB/GPU stream, %d injected single-bit faults/GPU/step"
                            % (self.bl, self.nb * self.bl / 2**30, len(self.faults)),
                "block_len": self.bl, "blocks_per_gpu": self.nb, "replicas": 3,
                "parallelism": "dp%d (independent blocks)" % world}

    def roofline(self, kern_ms):
        b = float(self.nb) * (self.bl + 2)
        t = kern_ms * 1e-3
        return {"bound": "hbm", "kernel": "crc16_stream_kernel<3,2,true>", "achieved": b / t * 1e-9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": b / t * 1e-9 / HBM_PEAK_GBS, "kernel_ms": kern_ms, "algorithmic_bytes": b}

    def cpu(self):
        return cpu_baseline_items("crc16")


class SHA256(Workload):
    metric = "protected msgs/sec + corrected-fault count, sha256 TMR"
    unit = "msgs/s"
    dtype = "u32"

    def __init__(self, a, eng, dev, rank, coast_amd):
        self.nm = a.batch or (1 << 22)
        g = torch.Generator(device=dev).manual_seed(256 + rank)
        self.msgs = torch.randint(0, 256, (self.nm, 64), dtype=torch.uint8, device=dev, generator=g)
        self.out = torch.empty((self.nm, 32), dtype=torch.uint8, device=dev)
        self.cfg = coast_amd.XmrConfig(coast_amd.TMR)
        self.eng = eng
        rng = np.random.default_rng(1 + rank)  # (msg, replica, site in {m[t], a..h, state}, t, bit)
        items = rng.choice(self.nm, a.faults, replace=False)
        rows = []
        for it in items:
            site = int(rng.choice([coast_amd.SITE_SHA_M, coast_amd.SITE_SHA_WV, coast_amd.SITE_SHA_STATE]))
            step = int(rng.integers(0, 3)) if site == coast_amd.SITE_SHA_STATE else int(rng.integers(0, 128))
            rows.append((int(it), int(rng.integers(0, 3)), site, step, int(rng.integers(0, 32)), int(rng.integers(0, 8))))
        self.faults = coast_amd.make_faults(rows)
        self.units_per_step = self.nm

    def launch(self):
        self.eng.sha256_batch(self.msgs, 64, out=self.out, cfg=self.cfg)

    def check(self):
        import hashlib

        m = self.msgs[:64].cpu().numpy()
        d = self.out[:64].cpu().numpy()
        return all(hashlib.sha256(m[i].tobytes()).digest() == d[i].tobytes() for i in range(64))

    def config(self, world):
        return {"workload": "sha256 %d x 64-byte messages TMR + on-device injector (%d faults/GPU/step)"
                            % (self.nm, len(self.faults)), "msgs_per_gpu": self.nm, "replicas": 3,
                "parallelism": "dp%d (independent messages)" % world}

    def roofline(self, kern_ms):
        ops = float(self.nm) * (1400 + 900)  # ~1400 VALU ops for the data block (64 rounds + 48 schedule words), ~900 for the
        # data-free padding block (rounds only)
        t = kern_ms * 1e-3
        return {"bound": "valu", "kernel": "sha256_fast_kernel<3,true>", "achieved": ops / t * 1e-12,
                "peak": VALU_LANE_OPS * 1e-12, "unit": "T lane-ops/s (algorithmic)", "frac": ops / t / VALU_LANE_OPS,
                "executed_frac": 3 * ops / t / VALU_LANE_OPS, "kernel_ms": kern_ms,
                "hbm_achieved_GBs": self.nm * 96 / t * 1e-9, "algorithmic_bytes": float(self.nm) * 96}

    def cpu(self):
        return cpu_baseline_items("sha256")


class AES(Workload):
    metric = "protected blocks/sec + detected-fault count, aes-128 ECB DWC"
    unit = "blocks/s"
    dtype = "u8"

    def __init__(self, a, eng, dev, rank, coast_amd):
        self.n = a.batch or (1 << 20)
        g = torch.Generator(device=dev).manual_seed(128 + rank)
        self.pt = torch.randint(0, 256, (self.n, 16), dtype=torch.uint8, device=dev, generator=g)
        self.key = torch.randint(0, 256, (self.n, 16), dtype=torch.uint8, device=dev, generator=g)
        self.st, self.k = self.pt.clone(), self.key.clone()
        self.cfg = coast_amd.XmrConfig(coast_amd.DWC)
        self.eng = eng
        rng = np.random.default_rng(3 + rank)
        items = rng.choice(self.n, a.faults, replace=False)
        self.faults = coast_amd.make_faults([(int(it), int(rng.integers(0, 2)), coast_amd.SITE_AES_STATE,
                                              int(rng.integers(0, 11)), int(rng.integers(0, 32)), int(rng.integers(0, 4)))
                                             for it in items])
        self.units_per_step = self.n
        self.dir = 0

    def launch(self):  # alternate encrypt / decrypt in place: the key buffer is consumed and restored (reference contract)
        self.eng.aes128_batch(self.st, self.k, self.dir, cfg=self.cfg)
        if self.dir == 0:
            self.k.copy_(self.key)
        self.dir ^= 1

    def check(self):
        return True

    def config(self, world):
        return {"workload": "aes-128 ECB %d blocks, per-block keys, DWC 2-way compare, alternating enc/dec "
                            "(%d faults/GPU/step)" % (self.n, len(self.faults)), "blocks_per_gpu": self.n,
                "replicas": 2, "parallelism": "dp%d (independent blocks)" % world}

    def roofline(self, kern_ms):
        t = kern_ms * 1e-3
        b = float(self.n) * 64
        return {"bound": "valu", "kernel": "aes128_enc_fast_kernel<2> (enc) / aes128_xmr_kernel<2> (dec)", "achieved": b / t * 1e-9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s (VALU/LDS-lookup bound; HBM shown for scale)", "frac": b / t * 1e-9 / HBM_PEAK_GBS,
                "kernel_ms": kern_ms, "algorithmic_bytes": b}

    def cpu(self):
        return cpu_baseline_items("aes")


class CacheTest(Workload):
    metric = "protected bytes/sec + corrected-fault count, cache_test (calc_sum) TMR scrub"
    unit = "GB/s"
    dtype = "i32"

    def __init__(self, a, eng, dev, rank, coast_amd):
        self.n = 600                      # data_array_elements, tests/cache_test/cacheTest.c:78
        self.na = a.batch or (1 << 22)    # 4 Mi arrays x 2400 B = 9.4 GiB per GPU
        self.arr = torch.arange(self.n, dtype=torch.int32, device=dev).repeat(self.na, 1).contiguous()
        self.cfg = coast_amd.XmrConfig(coast_amd.TMR)
        self.eng = eng
        rng = np.random.default_rng(11 + rank)
        items = rng.choice(self.na, a.faults, replace=False)
        self.faults = coast_amd.make_faults([(int(it), int(rng.integers(0, 3)), coast_amd.SITE_CT_SUM,
                                              int(rng.integers(0, self.n + 1)), int(rng.integers(0, 32))) for it in items])
        self.units_per_step = self.na * self.n * 4 * 1e-9  # GB
        self.sums = self.nerrs = None

    def launch(self):
        self.sums, self.nerrs = self.eng.cache_test_batch(self.arr, cfg=self.cfg)

    def check(self):
        return bool((self.sums == 179700).all()) and not bool(self.nerrs.any())  # generateGolden, cacheTest.c:88

    def config(self, world):
        return {"workload": "cache_test calc_sum, %d-int arrays TMR, %.1f GiB/GPU, %d injected single-bit faults/GPU/step"
                            % (self.n, self.na * self.n * 4 / 2**30, len(self.faults)),
                "array_elems": self.n, "arrays_per_gpu": self.na, "replicas": 3,
                "parallelism": "dp%d (independent arrays)" % world}

    def roofline(self, kern_ms):
        b = float(self.na) * (self.n * 4 + 8)
        t = kern_ms * 1e-3
        return {"bound": "hbm", "kernel": "cache_test_kernel<3>", "achieved": b / t * 1e-9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": b / t * 1e-9 / HBM_PEAK_GBS, "kernel_ms": kern_ms, "algorithmic_bytes": b}

    def cpu(self):
        return cpu_baseline_items("cache_test")


class ChSha(Workload):
    metric = "protected bytes/sec + corrected-fault count, CHStone sha TMR"
    unit = "GB/s"
    dtype = "u32"

    def __init__(self, a, eng, dev, rank, coast_amd):
        self.len = 16384                  # the benchmark's message: 2 x 8192 bytes, tests/chstone/sha/sha.h:59-60
        self.nm = a.batch or (1 << 18)    # 256 Ki messages = 4 GiB per GPU
        g = torch.Generator(device=dev).manual_seed(21 + rank)
        self.msgs = torch.randint(0, 256, (self.nm, self.len), dtype=torch.uint8, device=dev, generator=g)
        self.out = torch.empty((self.nm, 5), dtype=torch.int32, device=dev)
        self.cfg = coast_amd.XmrConfig(coast_amd.TMR)
        self.eng, self.ca = eng, coast_amd
        rng = np.random.default_rng(5 + rank)
        items = rng.choice(self.nm, a.faults, replace=False)
        self.faults = coast_amd.make_faults([(int(it), int(rng.integers(0, 3)), coast_amd.SITE_CHSHA_WV,
                                              int(rng.integers(0, 257 * 80)), int(rng.integers(0, 32)), int(rng.integers(0, 5)))
                                             for it in items])
        self.units_per_step = self.nm * self.len * 1e-9  # GB

    def launch(self):
        self.eng.chsha_batch(self.msgs, self.len, out=self.out, cfg=self.cfg)

    def check(self):
        ref = self.eng.chsha_batch(self.msgs[:1024], self.len, cfg=self.ca.XmrConfig(self.ca.UNPROTECTED))
        return bool(torch.equal(ref, self.out[:1024]))

    def config(self, world):
        return {"workload": "CHStone sha, %d-byte messages TMR, %.1f GiB/GPU, %d injected single-bit faults/GPU/step"
                            % (self.len, self.nm * self.len / 2**30, len(self.faults)),
                "msg_len": self.len, "msgs_per_gpu": self.nm, "replicas": 3,
                "parallelism": "dp%d (independent messages)" % world}

    def roofline(self, kern_ms):
        # per 64-byte block and lane: 80 rounds x (2 v_bitop3 + 1 v_xor + 2 v_alignbit + 2 v_add3) at their measured issue
        # costs (2.44 / 2.7 / 4.2 / 4.2 cycles per wave-instruction, profiles/microbench_r01.txt) = 1952 cycles per wave
        # for 21 messages x 64 bytes (TMR): 0.69 B/cycle/SIMD
        b = float(self.nm) * self.len
        t = kern_ms * 1e-3
        ceiling = 21 * 64 / 1952.0 * 1024 * 2.1e9
        return {"bound": "valu", "kernel": "chsha_kernel<3,false>", "achieved": b / t * 1e-9, "peak": ceiling * 1e-9,
                "unit": "GB/s (instruction-mix ceiling of the 80-round transform at 2.1 GHz)", "frac": b / t / ceiling,
                "kernel_ms": kern_ms, "algorithmic_bytes": b, "hbm_frac": b / t * 1e-9 / HBM_PEAK_GBS}

    def cpu(self):
        return cpu_baseline_items("chsha")


WORKLOADS = {"mm": MM, "crc16": CRC16, "sha256": SHA256, "aes": AES, "cache_test": CacheTest, "chsha": ChSha}


def pmc_traffic(workload, cfg):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/traffic.json)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        for key, rec in json.load(open(path)).items():  # records are keyed "<workload>" or "<workload>_<variant>"
            if key.split("_")[0] == workload and all(cfg.get(k) == v for k, v in rec.get("match", {}).items()):
                return rec["hbm_bytes_per_launch"], rec.get("source")
    except (OSError, ValueError):
        pass
    return None, None


def main():
    a = parse()
    if a.faults < 0:
        a.faults = 4096 if a.workload == "mm" else 1024
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    ndev = max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local % ndev)  # one process per GPU; the modulo only matters for single-GPU dry runs
        backend = os.environ.get("COAST_BENCH_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; gloo for 1-GPU dry runs
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local % ndev))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import coast_amd
    from coast_amd.dist import allreduce_counters

    eng = coast_amd.Engine(dev.index)
    wl = WORKLOADS[a.workload](a, eng, dev, rank, coast_amd)

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]

    def step(i=None):
        if len(wl.faults):
            eng.inject_faults(wl.faults)
        if i is not None:
            ev0[i].record()
        wl.launch()
        if i is not None:
            ev1[i].record()
        eng.reduce_counters()
        return allreduce_counters(eng, dist)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    eng.reset_stats()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        tot = step(i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    tot = [int(x) for x in tot.cpu().tolist()]
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in zip(ev0, ev1)]))
    outputs_ok = wl.check()  # untimed: voted output equals the unprotected / independent result

    if rank == 0:
        cfg = wl.config(world)
        roof = wl.roofline(kern_ms)
        traffic, src = pmc_traffic(a.workload, cfg)
        roof["traffic"] = traffic
        if src:
            roof["traffic_source"] = src
        out = {
            "metric": wl.metric, "value": float(world) * wl.units_per_step * a.steps / dt, "unit": wl.unit,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
            "config": cfg,
            "corrected_faults": tot[0], "dwc_detected": tot[2], "injected_faults": len(wl.faults) * a.steps * world,
            "sync_count": tot[1], "outputs_match_unprotected": outputs_ok,
            "roofline": roof,
        }
        if not a.no_cpu_baseline and world == 1:  # reported baseline, rank 0 at N=1 only
            out["cpu_baseline"] = wl.cpu()
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
