#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: protected elems/sec + corrected-fault count,
matrixMultiply TMR (configs[1]: 256x256 uint32, 3 replicas + vote), 1..8 GPUs of one node.

A step = one pass of the protected hot path over one batch of synthetic inputs that are already resident in HBM:
arm the on-device injector with a seeded fault list, run the protected kernel on this GPU's shard of independent
blocks, fold the fault counters and all-reduce them across GPUs (RCCL; the only collective of the path).  Blocks shard
across ranks with no data exchange, so scaling is weak (per-GPU batch fixed).

`python bench.py --gpus N` starts the N ranks ITSELF (one process per GPU under torch.distributed.run, backend nccl = RCCL)
when it was not already launched by torchrun; under torchrun it checks that WORLD_SIZE == N.

Default workload = mm (the metric BASELINE.json quotes).  After the headline measurement the default run also times
short legs of the other BASELINE configs and reports them under "extra" (each with its own roofline / cpu_baseline):
at N = 1 crc16 (256- and 255-byte blocks), sha256 4 Mi + injector, aes 1 Mi DWC and config 1 (mm 32x32 CPU-TMR); at
N > 1 the crc16 stream (8 GiB per GPU: N = 8 is the 64 GiB config).  `--workload X` runs one of them as the headline
instead (development); `--no-extra` skips the legs.

Prints ONE JSON line (rank 0).  The oracle is used only for the cpu_baseline legs.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLK = 2.4e9            # MI355X_MICROARCH.md: max clock 2400 MHz
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# Integer-MAC issue ceiling of the chip.  MI355X_MICROARCH.md gives the full-rate VALU figure (256 CU x 4 SIMD x 32 lanes
# x 2.4 GHz = 78.6 T lane-ops/s = 157.3 TFLOP/s / 2) but no integer-multiply rates, so the MAC peak is the measured
# issue rate of v_mad_u64_u32 -- the one-instruction 32-bit MAC the VALU kernel is built from -- at 8 waves/SIMD:
# 5.11 cycles per wave-instruction = 30.78 T lane-MAC/s (tools/valu_microbench, profiles/microbench_r01.txt).
MAC_PEAK = 30.78e12
# int8 MFMA: MI355X_MICROARCH.md lists no spec figure, "~2x bf16 rate" (bf16 ~2.5 PFLOP/s dense) and a ubench ceiling of
# >= 3944 TOPS; tools/mfma_probe reaches 4.2-4.3 POPS with v_mfma_i32_32x32x32_i8 (32 cycles/instruction/SIMD at the
# ~2.05 GHz the chip sustains under MFMA load).  Peak = 2 x 2.5e15; the measured ceiling is reported next to it.
I8_MFMA_PEAK = 5.0e15
I8_MFMA_UBENCH = 4.25e15
# ... on constants.  On RANDOM operand bytes (what the limbs of random matrices are) the same probe sustains 2.94-3.49 POP/s: the
# matrix core is power-limited and clocks down to 1.4-1.66 GHz (tools/mfma_probe2, profiles/microbench_r02.txt).
I8_MFMA_RANDOM = 3.4e15
# v_mfma_i32_16x16x64_i8, one wave per SIMD, 16 / 48 independent accumulators: 2.54-2.76 POP/s on constants and on random bytes
I8_MFMA_16X16_UBENCH = 2.755e15
# ... and with two waves per SIMD (24 accumulators each, mm_mfma_blk2_kernel's regime): 4.58 POP/s on constants, 3.71 on random bytes
I8_MFMA_16X16_UBENCH_2W = 3.711e15
N_SIMD = 256 * 4
# measured issue cost, cycles per wave-instruction at 8 waves/SIMD (profiles/microbench_r01.txt; v_bitop3 from the same probe,
# DESIGN.md section 4.2)
CYC = {"v_alignbit": 4.15, "v_bitop3": 2.44, "v_add3": 4.35, "v_add": 2.75, "v_lshr": 2.75}
# LDS: one ds_read_b32 / ds_read_u8 wave-instruction occupies the CU's LDS for >= 2 cycles (two 32-lane groups, one cycle
# each when conflict-free: MI355X_MICROARCH.md section LDS) -> 32 lane-lookups per clock per CU
LDS_LOOKUP_PEAK = 256 * 32 * CLK


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="mm", choices=["mm", "crc16", "sha256", "aes", "cache_test", "chsha"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU items per step (0 = the BASELINE config's size)")
    ap.add_argument("--side", type=int, default=256)
    ap.add_argument("--block-len", type=int, default=256, help="crc16 block length in bytes (the reference's maximum is 255)")
    ap.add_argument("--faults", type=int, default=-1,
                    help="single-bit flips injected per GPU per step (default: 4096 for mm, 1024 otherwise)")
    ap.add_argument("--profile-every", type=int, default=0,
                    help="bracket every n-th launch with HIP timing events (its time counted n times) instead of the workload's own period")
    ap.add_argument("--single-staging", action="store_true",
                    help="mm: run with COAST_F_SINGLE_STAGING -- the matrix-core kernel's global -> LDS staging loads not cloned (6 %% faster, lower "
                         "coverage of register upsets: docs/design/campaign.md); the headline is quoted on the cloned form, the library's default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline workload only")
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


def _ncpu():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline_mm(side, budget_s=8.0, threads_wall_s=3.0):
    """matrix_multiply on the host cores, the two CPU restatements side by side (the real `opt-7 -TMR` binary cannot be
    built here):
      * default-mode TMR (oracle/cpu_tmr_baseline.c: memory x3, loop-condition votes, -countErrors) -- what COAST emits by
        default and what the published 2.9x / 4.5x overheads describe; 1 core and all cores;
      * the -noMemReplication model (oracle/coast_oracle.c: one memory copy, one vote per stored element) -- the rule set
        the GPU engine instantiates, so this is the like-for-like line next to the GPU number.
    Bounded sample: as many side x side matrices as fit the budget."""
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
    ureps, du = _time_budget(lambda: orc.mm_plain(f, s), min(2.0, budget_s))  # unprotected arithmetic: the CPU TMR overhead
    nreps, dn = _time_budget(lambda: orc.mm_xmr(f[None], s[None], replicas=3), min(4.0, budget_s))
    # all host cores over independent matrices (the reference itself is single-threaded; this is the generous bound).
    # Visible CPUs can exceed what the container may use, so calibrate with one matrix per thread first.
    ncpu = _ncpu()
    w1 = orc.cpu_tmr_mm_threads(f, s, gold, ncpu, 1)
    per_thread = max(1, min(4096, int(threads_wall_s / max(w1, 1e-4))))
    wall = orc.cpu_tmr_mm_threads(f, s, gold, ncpu, per_thread)
    return {
        "value": reps * side * side / dt, "unit": "protected elems/s", "cores": 1, "kind": "port",
        "sample": "%d x (%dx%d uint32 matrix_multiply + checkGolden), default-mode TMR restatement "
                  "(memory x3, loop-condition votes, -countErrors), gcc -O3, %.1f s" % (reps, side, side, dt),
        "unprotected_elems_per_s": ureps * side * side / du,
        "tmr_overhead_x": (dt / reps) / (du / ureps),
        "nomemreplication_model": {"value": nreps * side * side / dn, "unit": "protected elems/s", "cores": 1,
                                   "sample": "%d x %dx%d, oracle -noMemReplication TMR model (one memory copy, one vote per "
                                             "stored element: the GPU engine's rule set), gcc -O3, %.1f s" % (nreps, side, side, dn)},
        "host_cpus": ncpu,
        "all_cores": {"value": ncpu * per_thread * side * side / wall if wall > 0 else None, "cores": ncpu,
                      "unit": "protected elems/s",
                      "sample": "%d threads x %d matrices, %.1f s wall" % (ncpu, per_thread, wall)},
    }


def cpu_baseline_items(kind, budget_s=6.0, block_len=256):
    """The oracle's replicated model (-noMemReplication schedule, the one the GPU instantiates) timed on one core."""
    from oracle import oracle as orc

    orc.build()
    rng = np.random.default_rng(0)
    if kind == "crc16":
        data = rng.integers(0, 256, (1 << 14, block_len), dtype=np.uint8)
        reps, dt = _time_budget(lambda: orc.crc16_xmr(data, block_len, replicas=3), budget_s)
        return {"value": reps * data.size / dt * 1e-9, "unit": "GB/s", "cores": 1, "kind": "port",
                "sample": "%d x %.1f MiB (16384 blocks x %d B), oracle TMR model, gcc -O3, %.1f s"
                          % (reps, data.size / 2**20, block_len, dt)}
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


def cpu_baseline_config1_mm32(eng, coast_amd):
    """BASELINE.json configs[0]: tests/matrixMultiply 32x32, TMR on the host CPU (no GPU in the measurement).  The reference
    LLVM pass cannot run here, so this is its default-mode restatement (1 core and all cores) plus the -noMemReplication
    model; TMR_ERROR_CNT is reported under a seeded list of K single-bit register upsets (one per run, the reference
    campaign's regime: threadFunctions.py:588-600).  The same list goes through the GPU engine once, as a cross-check of
    the count (not timed)."""
    import random

    from oracle import oracle as orc

    orc.build()
    n, K = 32, 1024
    random.seed(0)  # the reference generator's algorithm (tests/mm_common/mm_generator.py:42-51), seed 0, size 32:
    f = np.array([[random.randint(0, 2**32 - 1) for _ in range(n)] for _ in range(n)], dtype=np.uint32)
    s = np.array([[random.randint(0, 2**32 - 1) for _ in range(n)] for _ in range(n)], dtype=np.uint32)
    gold = orc.mm_xor(orc.mm_plain(f, s))
    assert gold == 1605501056
    out = cpu_baseline_mm(n, budget_s=3.0, threads_wall_s=2.0)
    rng = np.random.default_rng(32)
    rows = [(int(b * n * n + rng.integers(0, n * n)), int(rng.integers(0, 3)), int(rng.integers(0, 3)),
             int(rng.integers(0, n + 1)), int(rng.integers(0, 32))) for b in range(K)]
    fl = coast_amd.make_faults(rows)  # run b = matrix b of a batch of K identical matrices, one upset each
    fb, sb = np.repeat(f[None], K, 0), np.repeat(s[None], K, 0)
    t0 = time.perf_counter()
    r, st, det = orc.mm_xmr(fb, sb, replicas=3, faults=fl)
    dt = time.perf_counter() - t0
    runs_wrong = int(sum(int(np.bitwise_xor.reduce(r[b].reshape(-1))) != gold for b in range(K)))
    dm = orc.cpu_tmr_mm_campaign(f, s, gold, fl)  # default-mode restatement, one run per upset
    eng.reset_stats()
    eng.inject_faults(fl)
    g = eng.mm_batch(torch.from_numpy(fb.view(np.int32)).cuda(), torch.from_numpy(sb.view(np.int32)).cuda())
    gst = eng.stats()
    out.update({
        "workload": "matrixMultiply 32x32 uint32 (mm_generator.py seed 0, xor_golden 1605501056), TMR on the host CPU",
        "seeded_fault_list": {
            "runs": K, "faults_per_run": 1,
            "TMR_ERROR_CNT_nomemreplication_model": int(st["errors_corrected"]),
            "runs_with_wrong_golden": runs_wrong, "model_wall_s": dt,
            "default_mode": dm,
            "gpu_engine_same_list": {"TMR_ERROR_CNT": int(gst["errors_corrected"]), "sync_count": int(gst["sync_count"]),
                                     "outputs_equal_model": bool((g.cpu().numpy().view(np.uint32) == r).all())},
        },
    })
    return out


# ------------------------------------------------------------------------------------------------ workloads
class Workload:
    """setup() allocates device-resident synthetic inputs; launch() enqueues one protected pass."""
    kernels_per_step = 1
    # HIP timing events around every launch (1), or around every n-th one, its time counted n times (coast_set_profiling(ctx, n)): a pair
    # of event packets costs the stream ~15 us per step next to a 60 us kernel (profiles/r05_aes_step.txt).  Only the 1 Mi-block aes leg
    # samples (for >= 0.5 ms kernels the packets cost < 2 % and a sampled kernel_ms is no bound on the step).
    profile_every = 1

    def free(self):
        for k in list(self.__dict__):
            if isinstance(self.__dict__[k], torch.Tensor):
                del self.__dict__[k]
        torch.cuda.empty_cache()


class MM(Workload):
    name = "mm"
    metric = "protected elems/sec + corrected-fault count, matrixMultiply TMR"
    unit = "protected elems/s"
    dtype = "u32"

    def __init__(self, a, eng, dev, rank, coast_amd):
        self.n, self.batch = a.side, a.batch or 16384  # 12.9 GB of f, s, r: ~9 ms kernels (a batch, not one matrix)
        n, batch = self.n, self.batch
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        self.f = torch.randint(-2**31, 2**31, (batch, n, n), dtype=torch.int32, device=dev, generator=g)
        self.s = torch.randint(-2**31, 2**31, (batch, n, n), dtype=torch.int32, device=dev, generator=g)
        self.r = torch.empty_like(self.f)
        # The headline is quoted on the PROTECTED form -- the library's default since ABI 8: the
        # matrix-core kernel's global -> LDS staging loads cloned, compared, a third load deciding -- what the pass does to every load
        # (cloning.cpp:2187-2209, 2247-2255).  It costs 6 % kernel time in mm_mfma_blk4_kernel (profiles/r06_mm_blk4_ab.txt); the coverage of
        # single-bit register upsets with and without it: docs/design/campaign.md.  --single-staging / extra.mm_single_staging:
        # COAST_F_SINGLE_STAGING, one staging register set.
        self.clone = not bool(getattr(a, "single_staging", False)) and self.n == 256
        self.cfg = coast_amd.XmrConfig(coast_amd.TMR, 0, 0 if self.clone else coast_amd.F_SINGLE_STAGING)
        self.eng, self.ca = eng, coast_amd
        # one accumulator upset in one replica of K distinct output elements: each must be out-voted and counted once
        rng = np.random.default_rng(99 + rank)
        items = rng.choice(batch * n * n, a.faults, replace=False)
        self.fault_rows = [(int(it), int(rng.integers(0, 3)), coast_amd.SITE_MM_ACC, int(rng.integers(0, n + 1)),
                            int(rng.integers(0, 32))) for it in items]
        if getattr(a, "mm_phys", False):
            # the same number of upsets, but REAL ones: COAST_SITE_MM_VGPR flips of accumulator / B-fragment / A-fragment registers of the
            # running kernel (its PHYS instantiation) instead of the analytic delta of a logical accumulator upset
            self.fault_rows = []
            for it in items:
                reg = int(rng.integers(0, 12))  # 0-3 A fragment, 4-7 B fragment, 8-11 limb-sum accumulator
                slab = int(rng.integers(1, 4)) if reg >= 8 else int(rng.integers(0, 4))
                step = slab | (int(rng.integers(0, 64)) << 8) | (int(rng.integers(0, 4)) << 16) | (reg << 24)
                self.fault_rows.append((int(it), int(rng.integers(0, 3)), coast_amd.SITE_MM_VGPR, step, int(rng.integers(0, 32))))
        self.faults = coast_amd.make_faults(self.fault_rows)
        self.units_per_step = batch * n * n

    def launch(self):
        self.eng.mm_batch(self.f, self.s, out=self.r, cfg=self.cfg)

    def check(self):
        """Every element an upset was armed on, plus a random sample of 64 Ki others, against sum_k f[i][k] * s[k][j] mod 2^32
        computed by torch in int64 (wrap-around keeps the low word exact) -- independent of the engine's own kernels.  A vote
        that let an upset through, or stored the wrong copy, fails here."""
        n, nn = self.n, self.n * self.n
        rng = np.random.default_rng(4242)
        items = np.concatenate([np.array([int(f[0]) for f in self.fault_rows], dtype=np.int64),
                                rng.integers(0, self.batch * nn, 65536, dtype=np.int64)])
        it = torch.from_numpy(items).to(self.f.device)
        b, i, j = it // nn, (it % nn) // n, it % n
        ok = True
        for lo in range(0, it.numel(), 8192):
            sl = slice(lo, lo + 8192)
            fr = self.f[b[sl], i[sl], :].to(torch.int64) & 0xFFFFFFFF
            sc = self.s[b[sl], :, j[sl]].to(torch.int64) & 0xFFFFFFFF
            want = (fr * sc).sum(-1) & 0xFFFFFFFF
            got = self.r[b[sl], i[sl], j[sl]].to(torch.int64) & 0xFFFFFFFF
            ok = ok and bool(torch.equal(want, got))
        self.checked = {"faulted_elements": len(self.fault_rows), "sampled_elements": 65536, "reference": "torch int64 dot products"}
        return ok

    def config(self, world):
        cfg = {"workload": "matrixMultiply %dx%d uint32 TMR (3 replicas + vote%s), batch %d matrices/GPU, "
                           "%d injected single-bit faults/GPU/step" % (self.n, self.n, ", staging loads cloned" if self.clone else "", self.batch,
                                                                       len(self.faults)),
               "side": self.n, "batch_per_gpu": self.batch, "replicas": 3, "engine": self.engine(),
               "parallelism": "dp%d (independent matrices)" % world, "clone_staging": self.clone}
        if self.engine() == "mfma":
            cfg["tile"] = self.tile()  # where the replicas live on the matrix core (coast_hip.hip LAUNCH_MM)
        return cfg

    def engine(self):
        """which kernel coast_mm_batch dispatches to (coast_hip.hip LAUNCH_MM): side 256 runs on the matrix cores"""
        return "mfma" if self.n == 256 and os.environ.get("COAST_MM_ENGINE") != "valu" else "valu"

    @staticmethod
    def tile():
        """panel128 (default since round 6): mm_mfma_blk4_kernel (a workgroup owns 128 rows: two column-tile lanes x four row quarters, s
        converted twice per matrix; replica = accumulator block, in-lane vote, two waves per SIMD, every loaded operand replicated);
        blocks3: mm_mfma_blk3_kernel (the same on a 64-row panel, rounds 4-5's default); blocks2: mm_mfma_blk2_kernel (one A fragment set
        for the three replicas); lanes: mm_mfma_panel_kernel (the replicas in adjacent lanes, north_star's layout)"""
        t = os.environ.get("COAST_MM_TILE")
        return t if t in ("lanes", "blocks2", "blocks3") else "panel128"

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
            if self.tile() != "lanes":
                two = True
                kern = {"blocks3": "mm_mfma_blk3_kernel<3, false, 0, %s>" % ("true" if self.clone else "false"),
                        "panel128": "mm_mfma_blk4_kernel<false, %s>" % ("true" if self.clone else "false"),
                        "blocks2": "mm_mfma_blk2_kernel<3, false>"}[self.tile()]
                return dict(hbm, **{
                    "bound": "mfma", "kernel": kern,
                    "achieved": ops / t * 1e-12, "peak": I8_MFMA_PEAK * 1e-12, "unit": "TOP/s (int8)",
                    "frac": ops / t / I8_MFMA_PEAK,
                    # v_mfma_i32_16x16x64_i8 issued back to back by one wave per SIMD: 2.54-2.76 POP/s, constants or random
                    # bytes alike -- issue-bound, not power-bound; by two waves per SIMD: 3.71 POP/s on random bytes
                    # (tools/mfma_probe2 rate16x16x64, profiles/microbench_r02.txt).  Every executed MFMA is useful here
                    # (no lane padding, no ragged tile).
                    "frac_of_ubench_ceiling": ops / t / (I8_MFMA_16X16_UBENCH_2W if two else I8_MFMA_16X16_UBENCH),
                    "u32_macs_per_s": macs / t, "int8_ops_per_u32_mac": 2 * 10 * 3,
                    "algorithmic_macs_vs_valu_ceiling": macs / t / MAC_PEAK, "executed_macs_vs_valu_ceiling": 3.0 * macs / t / MAC_PEAK,
                    "note": "r = sum_k f*s mod 2^32 as ten int8 GEMMs of signed-byte limbs on v_mfma_i32_16x16x64_i8, the three replicas "
                            "in three accumulator blocks of the same lane, voted in-lane; "
                            + ("every replica's MFMAs read their own A and B fragments from LDS (the loads are replicated, the memory is "
                               "not: cloning.cpp:2187-2209, 2247-2255); " + ("the global -> LDS staging loads are cloned too and compared "
                               "in front of their first use (the default; + 6 % kernel time); " if self.clone else
                               "ONE staging register set on the way into LDS (COAST_F_SINGLE_STAGING); ") if self.tile() in ("blocks3", "panel128") else
                               "own B-operand registers and MFMAs per replica, ONE A fragment set for the three; ")
                            + ("two waves per SIMD, each with half the tile's rows (96 accumulator registers)" if two else
                               "one wave per SIMD (192 accumulator registers)") +
                            "; achieved = 2*N^3*10*3 int8 ops per matrix / kernel time; "
                            "peak = 2x the bf16 dense peak (MI355X_MICROARCH.md: I8 runs at ~2x bf16 rate)",
                })
            return dict(hbm, **{
                "bound": "mfma", "kernel": "mm_mfma_panel_kernel<3>",
                "achieved": ops / t * 1e-12, "peak": I8_MFMA_PEAK * 1e-12, "unit": "TOP/s (int8)",
                "frac": ops / t / I8_MFMA_PEAK, "frac_of_ubench_ceiling": ops / t / I8_MFMA_UBENCH,
                # executed int8 ops (32/30 lane-column padding, ragged 26th column tile: x 1.083) against what the matrix core
                # sustains on random operands at the chip's power limit
                "executed_frac_of_random_data_ceiling": 1.0833 * ops / t / I8_MFMA_RANDOM,
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


class MMDefaultMode(MM):
    """COAST's DEFAULT mode (docs/source/passes.rst:329, 337; cloning.cpp:2417-2537): memory is replicated too -- every replica multiplies
    its own copy of f and s into its own copy of r -- and what leaves the region is voted (and the copies repaired) at the exit.  Here: three
    unprotected launches on three memory images + coast_sync_copies over the three products.  HBM-bound: 3 x (read f, s; write r) + the
    vote's 3 reads and 1 write of r."""
    name = "mm_default_mode"
    metric = "protected elems/sec, matrixMultiply TMR, memory replicated (COAST default mode)"

    def __init__(self, a, eng, dev, rank, coast_amd):
        super().__init__(a, eng, dev, rank, coast_amd)
        self.fs = [(self.f, self.s)] + [(self.f.clone(), self.s.clone()) for _ in range(2)]
        self.rs = [torch.empty_like(self.f) for _ in range(3)]
        self.faults = coast_amd.make_faults([])  # (memory upsets of a copy: tools/campaign.py -s memory --mem-mode default)
        self.fault_rows = []
        self.clean = coast_amd.XmrConfig(coast_amd.UNPROTECTED)
        self.kernels_per_step = 4

    def launch(self):
        for (f, s), r in zip(self.fs, self.rs):
            self.eng.mm_batch(f, s, out=r, cfg=self.clean)
        self.eng.sync_copies(self.rs, out=self.r, scrub=True)

    def config(self, world):
        return {"workload": "matrixMultiply %dx%d uint32 TMR, memory replicated x3 (default mode), batch %d matrices/GPU" % (self.n, self.n, self.batch),
                "side": self.n, "batch_per_gpu": self.batch, "replicas": 3, "engine": "mfma x3 + exit vote",
                "parallelism": "dp%d (independent matrices)" % world}

    def roofline(self, kern_ms):
        n, batch = self.n, self.batch
        bytes_alg = float(batch) * n * n * 4 * (3 * 3 + 3 + 1)  # three launches x (f, s, r) + the vote: three products read, one written
        t = kern_ms * 1e-3
        return {"bound": "hbm", "kernel": "3 x mm_mfma_blk3_kernel<1, false> + sync_copies_kernel", "kernel_ms": kern_ms,
                "achieved": bytes_alg / t * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_alg / t * 1e-9 / HBM_PEAK_GBS,
                "algorithmic_bytes": bytes_alg,
                "note": "kernel_ms = the four kernels of a step together (HIP events of the C ABI); algorithmic bytes = 3 x 12 n^2 (each replica's "
                        "own f, s, r) + 16 n^2 (exit vote: three products read, the voted one written; repairs of a disagreeing copy are not counted)"}


class CRC16(Workload):
    name = "crc16"
    metric = "protected bytes/sec + corrected-fault count, crc16 TMR stream"
    unit = "GB/s"
    dtype = "u16"

    def __init__(self, a, eng, dev, rank, coast_amd):
        self.bl = a.block_len
        self.nb = a.batch or (1 << 25)  # 2^25 blocks x 256 B = 8 GiB per GPU (64 GiB over 8 GPUs)
        g = torch.Generator(device=dev).manual_seed(16 + rank)
        off = int(os.environ.get("COAST_BENCH_CRC_OFFSET", "0"))  # development: start the stream `off` bytes into the allocation
        self.data = torch.empty(self.nb * self.bl + off, dtype=torch.uint8, device=dev)[off:]
        step = 1 << 28
        for off in range(0, self.data.numel(), step):  # generated on-device, shard by shard
            self.data[off:off + step].copy_(torch.randint(0, 256, (min(step, self.data.numel() - off),),
                                                          dtype=torch.uint8, device=dev, generator=g))
        self.out = torch.empty(self.nb, dtype=torch.int16, device=dev)
        self.cfg = coast_amd.XmrConfig(coast_amd.TMR)
        self.eng = eng
        rng = np.random.default_rng(7 + rank)
        items = rng.choice(self.nb, a.faults, replace=False)
        self.fault_rows = [(int(it), int(rng.integers(0, 3)), coast_amd.SITE_CRC_CRC, int(rng.integers(0, self.bl + 1)),
                            int(rng.integers(0, 16))) for it in items]
        self.faults = coast_amd.make_faults(self.fault_rows)
        self.units_per_step = self.nb * self.bl * 1e-9  # GB

    def launch(self):
        self.eng.crc16_batch(self.data, self.bl, out=self.out, cfg=self.cfg)

    def check(self):
        """every block an upset was armed on, the first 4096 and the last 64 blocks of the stream, against the byte-serial
        recurrence of tests/crc16/crc16.c:25-29 evaluated with numpy on the host -- independent of the engine's kernels"""
        idx = np.unique(np.concatenate([np.array([int(f[0]) for f in self.fault_rows], dtype=np.int64), np.arange(4096),
                                        np.arange(self.nb - 64, self.nb)]))
        idx = idx[(idx >= 0) & (idx < self.nb)]
        it = torch.from_numpy(idx).to(self.data.device)
        rows = self.data.view(-1)[(it[:, None] * self.bl + torch.arange(self.bl, device=self.data.device)[None, :])].cpu().numpy()
        crc = np.full(rows.shape[0], 0xFFFF, dtype=np.uint32)
        for t in range(self.bl):
            x = ((crc >> 8) ^ rows[:, t]) & 0xFF
            x ^= x >> 4
            crc = ((crc << 8) ^ (x << 12) ^ (x << 5) ^ x) & 0xFFFF
        got = self.out[it].cpu().numpy().view(np.uint16).astype(np.uint32)
        self.checked = {"faulted_blocks": len(self.fault_rows), "other_blocks": int(rows.shape[0] - len(self.fault_rows)),
                        "reference": "numpy byte-serial crc16"}
        return bool((got == crc).all())

    def config(self, world):
        return {"workload": "crc16 %d-byte blocks TMR, %.1f GiB/GPU stream, %d injected single-bit faults/GPU/step"
                            % (self.bl, self.nb * self.bl / 2**30, len(self.faults)),
                "block_len": self.bl, "blocks_per_gpu": self.nb, "replicas": 3,
                "parallelism": "dp%d (independent blocks)" % world}

    def roofline(self, kern_ms):
        b = float(self.nb) * (self.bl + 2)
        t = kern_ms * 1e-3
        return {"bound": "hbm", "kernel": "crc16_stream_kernel<3>", "achieved": b / t * 1e-9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": b / t * 1e-9 / HBM_PEAK_GBS, "kernel_ms": kern_ms, "algorithmic_bytes": b}

    def cpu(self):
        return cpu_baseline_items("crc16", block_len=self.bl)


class SHA256(Workload):
    name = "sha256"
    metric = "protected msgs/sec + corrected-fault count, sha256 TMR"
    unit = "msgs/s"
    dtype = "u32"
    # VALU instructions of one 64-byte message per replica lane (FIPS 180-4, as sha256_fast_kernel issues them; tools/instr_mix.py
    # counts the same numbers in the compiled kernel): two compressions (the data block and the padding block, whose schedule
    # is expanded per replica lane like any other -- it sits inside the triplicated sha256_transform), each 64 rounds + 48
    # schedule words.  round: 6 v_alignbit (the rotates of Sigma0/Sigma1) + 4 v_bitop3 (two xor3, ch, maj) + 3 v_add3 + 1 v_add;
    # schedule word: 4 v_alignbit + 2 v_lshr + 2 v_bitop3 + 2 v_add; 2 x 8 state adds.
    MIX = {"v_alignbit": 2 * (64 * 6 + 48 * 4), "v_bitop3": 2 * (64 * 4 + 48 * 2), "v_add3": 2 * 64 * 3,
           "v_add": 2 * (64 + 48 * 2) + 16, "v_lshr": 2 * 48 * 2}

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
        self.fault_rows = rows
        self.faults = coast_amd.make_faults(rows)
        self.units_per_step = self.nm

    def launch(self):
        self.eng.sha256_batch(self.msgs, 64, out=self.out, cfg=self.cfg)

    def check(self):
        import hashlib

        idx = np.unique(np.concatenate([np.array([int(f[0]) for f in self.fault_rows], dtype=np.int64), np.arange(256),
                                        np.arange(self.nm - 64, self.nm)]))
        it = torch.from_numpy(idx).to(self.msgs.device)
        m = self.msgs[it].cpu().numpy()
        d = self.out[it].cpu().numpy()
        self.checked = {"faulted_messages": len(self.fault_rows), "other_messages": int(len(idx) - len(self.fault_rows)),
                        "reference": "hashlib.sha256"}
        return all(hashlib.sha256(m[i].tobytes()).digest() == d[i].tobytes() for i in range(len(idx)))

    def config(self, world):
        return {"workload": "sha256 %d x 64-byte messages TMR + on-device injector (%d faults/GPU/step)"
                            % (self.nm, len(self.faults)), "msgs_per_gpu": self.nm, "replicas": 3,
                "parallelism": "dp%d (independent messages)" % world}

    def roofline(self, kern_ms):
        # VALU issue bound.  A TMR wave carries 21 messages; its instruction stream costs sum(count x measured issue cycles).
        cyc = sum(self.MIX[k] * CYC[k] for k in self.MIX)
        ceiling = N_SIMD * CLK / cyc * 21
        t = kern_ms * 1e-3
        return {"bound": "valu", "kernel": "sha256_fast_kernel<3,true>", "achieved": self.nm / t * 1e-9,
                "peak": ceiling * 1e-9, "unit": "G msgs/s (instruction-mix issue ceiling at 2.4 GHz)", "frac": self.nm / t / ceiling,
                "cycles_per_wave_of_21_msgs": cyc, "instruction_mix": self.MIX, "issue_cycles": CYC, "kernel_ms": kern_ms,
                # the same count against the guide's neutral full-rate VALU figure (256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T
                # lane-ops/s, MI355X_MICROARCH.md): 3 replica lanes x sum(MIX) instructions per message
                "valu_lane_ops_per_s": self.nm / t * 3 * sum(self.MIX.values()),
                "frac_of_fullrate_valu": self.nm / t * 3 * sum(self.MIX.values()) / (N_SIMD * 32 * CLK),
                "hbm_achieved_GBs": self.nm * 96 / t * 1e-9, "algorithmic_bytes": float(self.nm) * 96}

    def cpu(self):
        return cpu_baseline_items("sha256")


class AES(Workload):
    name = "aes"
    metric = "protected blocks/sec + detected-fault count, aes-128 ECB DWC"
    unit = "blocks/s"
    dtype = "u8"
    kernels_per_step = 1
    # LDS lookups of one block per replica lane (tools/instr_mix.py on the compiled kernels).  One-copy tables: encryption 147
    # ds_read_b32 (four T-tables x 4 columns x 9 rounds + last round) + 56 ds_read_u8 (S-box: key schedule, last round);
    # decryption 179 + 84.  Bank-replicated tables (what 1 Mi DWC blocks run): encryption 203 ds_read_b32 (the S-box bytes come
    # out of Te_0 entries), decryption 208 ds_read_b32 (144 Td, 48 S-box, 16 rsbox) + 32 ds_read_b64 ({Tis_0, S} pairs: one lookup per
    # key-schedule byte of the main rounds)
    LOOKUPS = {0: 203, 1: 208 + 32}

    def __init__(self, a, eng, dev, rank, coast_amd):
        self.n = a.batch or (1 << 20)
        g = torch.Generator(device=dev).manual_seed(128 + rank)
        self.pt = torch.randint(0, 256, (self.n, 16), dtype=torch.uint8, device=dev, generator=g)
        self.key = torch.randint(0, 256, (self.n, 16), dtype=torch.uint8, device=dev, generator=g)
        self.st, self.k = self.pt.clone(), self.key.clone()
        self.cfg = coast_amd.XmrConfig(coast_amd.DWC)
        self.eng, self.ca = eng, coast_amd
        rng = np.random.default_rng(3 + rank)
        items = rng.choice(self.n, a.faults, replace=False)
        self.faults = coast_amd.make_faults([(int(it), int(rng.integers(0, 2)), coast_amd.SITE_AES_STATE,
                                              int(rng.integers(0, 11)), int(rng.integers(0, 32)), int(rng.integers(0, 4)))
                                             for it in items])
        self.units_per_step = self.n
        self.dir = 0
        # a 50-70 us launch: bracket every 5th one (odd: the steps alternate encrypt / decrypt, both directions are sampled alike)
        self.profile_every = 5 if self.n <= (1 << 21) else 1

    def launch(self):
        """alternate encrypt / decrypt in place, ONE launch per step.  The reference contract (TI_aes_128.c:107-231): encryption leaves
        the last round key in the key buffer, decryption takes a cipher key and ends on it -- so the buffer is never restored by the
        harness: the decrypt step runs with the previous step's last round keys as its cipher keys (any 16 bytes are a key), the next
        encrypt step with those again.  decrypt(encrypt(x)) == x under one key is check()'s business, on a sample."""
        self.eng.aes128_batch(self.st, self.k, self.dir, cfg=self.cfg)
        self.dir ^= 1

    def check(self):
        """clean sample: DWC result == unprotected result, decrypt(encrypt(x)) == x, and the in-place key contract
        (encrypt leaves the last round key, decrypt walks it back to the cipher key; TI_aes_128.c:107-231)"""
        ca = self.ca
        m = min(self.n, 4096)
        pt, key = self.pt[:m].clone(), self.key[:m].clone()
        s1, k1 = pt.clone(), key.clone()
        self.eng.aes128_batch(s1, k1, 0, cfg=ca.XmrConfig(ca.DWC))
        s2, k2 = pt.clone(), key.clone()
        self.eng.aes128_batch(s2, k2, 0, cfg=ca.XmrConfig(ca.UNPROTECTED))
        ok = torch.equal(s1, s2) and torch.equal(k1, k2) and not torch.equal(s1, pt)
        k3 = key.clone()
        self.eng.aes128_batch(s1, k3, 1, cfg=ca.XmrConfig(ca.DWC))
        return bool(ok and torch.equal(s1, pt) and torch.equal(k3, key))

    def config(self, world):
        return {"workload": "aes-128 ECB %d blocks, per-block keys, DWC 2-way compare, alternating enc/dec in place "
                            "(%d faults/GPU/step)" % (self.n, len(self.faults)), "blocks_per_gpu": self.n,
                "replicas": 2, "parallelism": "dp%d (independent blocks)" % world}

    def roofline(self, kern_ms):
        t = kern_ms * 1e-3
        look = 0.5 * (self.LOOKUPS[0] + self.LOOKUPS[1]) * 2  # per block: mean of the enc / dec steps, x 2 replica lanes
        return {"bound": "lds", "kernel": "aes128_enc_rep_kernel<2> / aes128_dec_rep_kernel<2> (alternating)",
                "achieved": self.n * look / t * 1e-12, "peak": LDS_LOOKUP_PEAK * 1e-12,
                "unit": "T lane-lookups/s (LDS: 32 conflict-free 4-byte lookups per clock per CU at 2.4 GHz)",
                "frac": self.n * look / t / LDS_LOOKUP_PEAK, "lookups_per_block_lane": self.LOOKUPS, "kernel_ms": kern_ms,
                "hbm_achieved_GBs": self.n * 64 / t * 1e-9, "algorithmic_bytes": float(self.n) * 64,
                "note": "bank-replicated tables: the lookups are conflict-free and their address is one v_perm_b32; the kernels are "
                        "bound by their VALU instruction count (466 / 717 per block and lane); 1 Mi blocks are a 46-64 us launch clean, the "
                        "armed upsets are applied inside it (+ ~12 us: profiles/r04_aes_step.txt); one launch per step, no harness copy"
                        + ("; the counter fold runs in the kernel's exit path (the last workgroup out), kernel_ms contains it and no fold kernel "
                           "follows the launch (profiles/r05_aes_step.txt; COAST_AES_FOLD=0 takes it out)"
                           if os.environ.get("COAST_AES_FOLD", "") != "0" else "; COAST_AES_FOLD=0: a separate fold kernel behind every launch")}

    def cpu(self):
        return cpu_baseline_items("aes")


class CacheTest(Workload):
    name = "cache_test"
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
    name = "chsha"
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
            if (key == workload or key.startswith(workload + "_")) and all(cfg.get(k) == v for k, v in rec.get("match", {}).items()):
                return rec["hbm_bytes_per_launch"], rec.get("source")
    except (OSError, ValueError):
        pass
    return None, None


# ------------------------------------------------------------------------------------------------ one timed run
def timed_run(wl, eng, dist, dev, steps, warmup, world):
    """W untimed + K timed steps of one workload, barrier + synchronize on both sides, MAX over ranks.  Kernel time comes
    from HIP events the C ABI records around every protected launch on the launch stream (coast_stats.kernel_ms)."""
    from coast_amd.dist import allreduce_counters

    def step():
        if len(wl.faults):
            eng.inject_faults(wl.faults)
        wl.launch()
        eng.reduce_counters()
        return allreduce_counters(eng, dist, snapshot=False)  # (one rank, no process group: the live totals, no copy kernel)

    # sampled brackets only where at least four of them fit the timed region AND the timed launches are a whole number of periods (each
    # bracket stands for `every` launches: ADVICE r5 -- 22 steps at every = 5 would be 20 launches of time divided by 22)
    every = wl.profile_every if steps >= 4 * wl.profile_every and steps % wl.profile_every == 0 else 1
    eng.set_profiling(every)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    eng.set_profiling(every)  # (restarts the period: the first timed launch is a bracketed one whatever the warm-up count was)
    eng.reset_stats()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        tot = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    my_dt = dt
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    tot = [int(x) for x in tot.cpu().tolist()]
    st = eng.stats()
    kern_ms = st["kernel_ms"] / max(steps, 1)
    eng.set_profiling(1)
    out = {"dt": dt, "totals": tot, "kernel_ms": kern_ms, "launch_info": eng.last_launch(), "profile_every": every,
           "hbm_bytes_per_step": st["hbm_bytes"] / max(steps, 1)}
    if dist:
        out["ranks"] = rank_diagnostics(eng, dist, dev, kern_ms, my_dt / steps * 1e3)
    return out


def rank_diagnostics(eng, dist, dev, kern_ms, step_ms, reps=20):
    """What an N-rank line needs to be read: every rank's kernel time and own step time (the line's ms_per_step is their maximum), and
    the latency of the one collective -- the all-reduce of the four fault counters -- timed by itself, AFTER the timed region: HIP
    events on the current stream around the RCCL call (a synchronous torch.distributed op makes the current stream wait for RCCL's)."""
    from coast_amd.dist import allreduce_counters

    on_host = dist.get_backend() == "gloo"
    coll_us = None
    if not on_host:
        for _ in range(3):
            allreduce_counters(eng, dist)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = eng.counters.clone()
        e0.record()
        for _ in range(reps):
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        e1.record()
        torch.cuda.synchronize()
        coll_us = e0.elapsed_time(e1) / reps * 1e3
    else:  # dry run of the rank logic without RCCL: wall clock around the host-staged all-reduce
        t0 = time.perf_counter()
        for _ in range(reps):
            allreduce_counters(eng, dist)
        coll_us = (time.perf_counter() - t0) / reps * 1e6
    mine = torch.tensor([float(dist.get_rank()), kern_ms, step_ms, coll_us], dtype=torch.float64, device="cpu" if on_host else dev)
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    rows = [[float(x) for x in t.cpu().tolist()] for t in every]
    return {"per_rank": [{"rank": int(r[0]), "kernel_ms": r[1], "step_ms": r[2], "collective_us": r[3]} for r in rows],
            "slowest_rank": int(max(rows, key=lambda r: r[2])[0]),
            "collective_us": max(r[3] for r in rows),
            "collective_timing": ("HIP events around %d back-to-back RCCL all_reduce(SUM) of 4 x int64 after the timed region" % reps)
                                 if not on_host else "wall clock around the gloo (host-staged) all_reduce: a dry run, not xGMI"}


def result_fields(wl, run, a, world, steps, warmup, with_cpu):
    cfg = wl.config(world)
    roof = wl.roofline(run["kernel_ms"])
    traffic, src = pmc_traffic(wl.name, cfg)
    roof["traffic"] = traffic
    pe = run.get("profile_every", 1)
    roof["kernel_ms_from"] = ("HIP events around every launch of the timed region" if pe == 1 else
                              "HIP events around every %dth launch of the timed region, each counted %d times" % (pe, pe))
    roof["kernel_ms_sampled"] = pe != 1
    step_ms = run["dt"] / steps * 1e3
    # the dominant kernel cannot take longer than the step that contains it: true by construction when every launch is bracketed
    kernel_le_step = run["kernel_ms"] <= step_ms * 1.001
    if not kernel_le_step:
        sys.stderr.write("bench.py: %s: kernel_ms %.4f > ms_per_step %.4f (%s)\n"
                         % (wl.name, run["kernel_ms"], step_ms, "sampled estimate" if pe != 1 else "TIMING INCONSISTENT"))
        assert pe != 1, "kernel_ms > ms_per_step with every launch bracketed"
    if src:
        # a committed measurement of this same command, looked up by configuration -- NOT a counter pass of this very run (PMC
        # collection serialises the kernels and needs rocprofv3 around the process: tools/profile.sh)
        roof["traffic_kind"] = "static: profiles/traffic.json"
        roof["traffic_source"] = src
    tot = run["totals"]
    out = {
        "metric": wl.metric, "value": float(world) * wl.units_per_step * steps / run["dt"], "unit": wl.unit,
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": run["dt"] / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
        "config": cfg,
        "corrected_faults": tot[0], "dwc_detected": tot[2], "injected_faults": len(wl.faults) * steps * world,
        "sync_count": tot[1], "outputs_match_unprotected": wl.check(),
        "voted_by": run["launch_info"]["engine"], "stepwise_blocks_last_launch": run["launch_info"]["general_blocks"],
        # tiles / workgroups of the lean kernel that owned an armed upset and applied, voted and counted it themselves
        "hooked_blocks_last_launch": run["launch_info"]["hooked_blocks"],
        "roofline": roof, "kernel_le_step": kernel_le_step,
        # what a timed step contains (not like-for-like with rounds 2-3, whose steps uploaded the table every time)
        "timed_step": "inject (the step's upset table equals the previous step's: it stays resident on the device -- a host compare, no upload; the "
                      "kernel still applies every upset) + launch + fold of the per-workgroup counter slots into the totals",
    }
    if getattr(wl, "checked", None):
        out["outputs_checked"] = wl.checked
    if run.get("ranks"):
        out["ranks"] = run["ranks"]
        if "algorithmic_bytes" in roof:  # every GPU's own fraction of the HBM roofline (the 8-GPU crc16 stream is priced per GPU)
            for r in out["ranks"]["per_rank"]:
                r["hbm_frac"] = roof["algorithmic_bytes"] / (r["kernel_ms"] * 1e-3) * 1e-9 / HBM_PEAK_GBS if r["kernel_ms"] > 0 else None
    if with_cpu:
        out["cpu_baseline"] = wl.cpu()
    return out


def extra_legs(a, eng, dist, dev, rank, world, coast_amd):
    """Short legs of the other BASELINE configs (same harness, same contract, fewer steps)."""
    import copy

    legs = {}
    # (both crc16 legs at every N: 255 bytes is the reference's own block length -- `unsigned char length` -- and the shape north_star's
    # 8-GPU stream has to hold its 40 % of the HBM roofline in)
    plan = [("crc16_256B", CRC16, {"block_len": 256}, 20, 5), ("crc16_255B", CRC16, {"block_len": 255}, 20, 5)]
    if world == 1:
        plan += [("sha256", SHA256, {}, 20, 5), ("aes", AES, {}, 60, 10),
                 # the BASELINE batch (1 Mi blocks) is a 50-80 us launch; the same kernels on 16 Mi blocks show what they sustain
                 ("aes_16Mi_blocks", AES, {"batch": 1 << 24}, 12, 4)]
    for name, cls, over, steps, warm in plan:
        torch.cuda.synchronize()
        time.sleep(1.0)  # the matrix-core leg leaves the chip at its power limit: let the clocks settle before an HBM-bound leg
        b = copy.copy(a)
        b.batch, b.faults = 0, 1024
        for k, v in over.items():
            setattr(b, k, v)
        wl = cls(b, eng, dev, rank, coast_amd)
        run = timed_run(wl, eng, dist, dev, steps, warm, world)
        if rank == 0:
            legs[name] = result_fields(wl, run, b, world, steps, warm, with_cpu=(world == 1 and not a.no_cpu_baseline))
        wl.free()
    if world == 1:
        # the headline's variants: one staging register set (COAST_F_SINGLE_STAGING: no clones), the armed
        # upsets as REAL register flips (the kernel's PHYS instantiation), north_star's replica layout (three adjacent lanes, cross-lane voter:
        # COAST_MM_TILE=lanes), and COAST's default mode (memory replicated)
        for name, cls, over, env in (("mm_single_staging", MM, {"batch": 8192, "single_staging": True}, {}),
                                     # rounds 4-5's kernel (64-row panel) in the form the round-5 line was quoted on
                                     ("mm_blocks3_clones", MM, {"batch": 8192}, {"COAST_MM_TILE": "blocks3"}),
                                     ("mm_physical_upsets", MM, {"mm_phys": True, "batch": 8192, "single_staging": True}, {}),
                                     ("mm_lane_replicas", MM, {"batch": 8192, "single_staging": True}, {"COAST_MM_TILE": "lanes"}),
                                     ("mm_default_mode", MMDefaultMode, {"batch": 8192}, {})):
            torch.cuda.synchronize()
            time.sleep(1.0)
            b = copy.copy(a)
            b.batch, b.faults = 0, 4096
            for k, v in over.items():
                setattr(b, k, v)
            saved = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                wl = cls(b, eng, dev, rank, coast_amd)
                run = timed_run(wl, eng, dist, dev, 10, 3, world)
                legs[name] = result_fields(wl, run, b, world, 10, 3, with_cpu=False)
                wl.free()
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
    if world == 1 and rank == 0 and not a.no_cpu_baseline:
        legs["config1_mm32_cpu_tmr"] = cpu_baseline_config1_mm32(eng, coast_amd)
    return legs


# ------------------------------------------------------------------------------------------------ the ONE stdout line
LINE_LIMIT = 6000  # the driver keeps only the tail of stdout (BENCH_r05: an 26 KB line -> "parsed": null); r04's record held ~8 KB
ROOF_KEYS = ("bound", "kernel", "kernel_ms", "kernel_ms_sampled", "achieved", "peak", "unit", "frac", "traffic", "traffic_kind",
             "algorithmic_bytes", "hbm_frac")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "host_cpus", "tmr_overhead_x")
TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "corrected_faults", "dwc_detected", "injected_faults", "sync_count", "outputs_match_unprotected",
            "kernel_le_step", "collective")


def _rnd(x, digits=6):
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    return x


def compact_result(out):
    """The fixed, short record the driver parses (the reference's own result record is a fixed tuple: decoder.py:66-86,
    sha256_tmr.c:30): the contract's keys, `roofline` and `cpu_baseline` without prose, one short row per extra leg."""
    line = {k: _rnd(out[k], 9) for k in TOP_KEYS if k in out}
    if isinstance(line.get("config"), dict):
        line["config"] = {k: v for k, v in line["config"].items() if k in ("workload", "side", "batch_per_gpu", "replicas", "engine", "tile",
                                                                            "parallelism", "clone_staging", "block_len", "blocks_per_gpu",
                                                                            "msgs_per_gpu")}
    line["roofline"] = {k: _rnd(out["roofline"][k]) for k in ROOF_KEYS if k in out.get("roofline", {})}
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = {k: _rnd(cb[k]) for k in CPU_KEYS if k in cb}
        if isinstance(cb.get("all_cores"), dict):
            line["cpu_baseline"]["all_cores_value"] = _rnd(cb["all_cores"].get("value"))
    if out.get("ranks"):
        rk = out["ranks"]
        line["ranks"] = {"slowest_rank": rk.get("slowest_rank"), "collective_us": _rnd(rk.get("collective_us"), 4),
                         "per_rank": [[r["rank"], _rnd(r["kernel_ms"], 5), _rnd(r["step_ms"], 5), _rnd(r.get("hbm_frac"), 4)]
                                      for r in rk.get("per_rank", [])],
                         "per_rank_cols": "rank, kernel_ms, step_ms, hbm_frac"}
    legs = out.get("extra") or {}
    if legs:
        # [value, unit, ms_per_step, kernel_ms, bound, frac, cpu_baseline value] per leg
        line["extra_cols"] = "value, unit, ms_per_step, kernel_ms, bound, frac, faults_counted, outputs_ok, cpu_value"
        line["extra_summary"] = {
            name: [_rnd(leg.get("value"), 5), leg.get("unit"), _rnd(leg.get("ms_per_step"), 5), _rnd(leg.get("roofline", {}).get("kernel_ms"), 5),
                   leg.get("roofline", {}).get("bound"), _rnd(leg.get("roofline", {}).get("frac"), 4),
                   (leg.get("corrected_faults", 0) or 0) + (leg.get("dwc_detected", 0) or 0) if "corrected_faults" in leg else None,
                   leg.get("outputs_match_unprotected"), _rnd((leg.get("cpu_baseline") or {}).get("value"), 4)]
            for name, leg in legs.items()}
    if out.get("full_record"):
        line["full_record"] = out["full_record"]
    return line


def final_line(out):
    """json of compact_result(out), guaranteed to fit LINE_LIMIT: legs are dropped from the END of extra_summary before anything else."""
    line = compact_result(out)
    text = json.dumps(line, separators=(",", ":"))
    while len(text) > LINE_LIMIT and line.get("extra_summary"):
        line["extra_summary"].popitem()
        line["extra_summary_truncated"] = True
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:
        for k in ("ranks", "cpu_baseline"):
            if len(text) > LINE_LIMIT and isinstance(line.get(k), dict) and "sample" in line[k]:
                line[k]["sample"] = line[k]["sample"][:80]
                text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def write_full_record(out):
    """Every leg with its prose (notes, samples, instruction mixes) goes to a FILE, never to stdout: $COAST_BENCH_FULL, else
    gpurun_out/bench_full.json when that scratch directory exists, else ./bench_full.json."""
    path = os.environ.get("COAST_BENCH_FULL")
    if not path:
        d = os.path.join(ROOT, "gpurun_out")
        path = os.path.join(d if os.path.isdir(d) else ROOT, "bench_full.json")
    try:
        with open(path, "w") as fh:
            json.dump(out, fh)
            fh.write("\n")
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def spawn(a):
    """--gpus N without a torchrun environment: start the N ranks here (one process per GPU, 127.0.0.1 rendezvous)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.faults < 0:
        a.faults = 4096 if a.workload == "mm" else 1024
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("COAST_BENCH_ECHO_RANK"):
        sys.stderr.write("bench.py: rank %d/%d (local %d)\n" % (rank, world, local))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or let bench.py spawn the ranks)"
                 % (a.gpus, world, a.gpus))
    dist = None
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit("bench.py: no GPU visible (the engine has no CPU path)")
    backend = os.environ.get("COAST_BENCH_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; gloo only for 1-GPU dry runs of N > 1
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl" and ndev < world:
            sys.exit("bench.py: --gpus %d needs %d GPUs, %d visible (RCCL wants one GPU per rank; COAST_BENCH_BACKEND=gloo "
                     "dry-runs the rank logic on fewer)" % (world, world, ndev))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local % ndev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local % ndev))
        else:
            dist.init_process_group(backend)
    elif os.environ.get("COAST_BENCH_FORCE_DIST"):
        # one rank, but through the multi-GPU code path: init_process_group("nccl", device_id=...), the device-tensor all_reduce of
        # the fault counters and the barriers all run on RCCL (a 1-GPU box can execute what the 8-GPU run executes per rank)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        torch.cuda.set_device(0)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group(backend, rank=0, world_size=1)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import coast_amd

    eng = coast_amd.Engine(dev.index)
    eng.set_profiling(True)
    wl = WORKLOADS[a.workload](a, eng, dev, rank, coast_amd)
    if a.profile_every > 0:
        wl.profile_every = a.profile_every
    run = timed_run(wl, eng, dist, dev, a.steps, a.warmup, world)
    out = None
    if rank == 0:
        out = result_fields(wl, run, a, world, a.steps, a.warmup, with_cpu=(world == 1 and not a.no_cpu_baseline))
        out["collective"] = ("%s all_reduce(SUM) of 4 x int64 fault counters per step, %d rank%s" % (backend, world, "s" if world > 1 else "")) \
            if dist else "none (1 rank)"
    if a.workload == "mm" and not a.no_extra:
        wl.free()
        legs = extra_legs(a, eng, dist, dev, rank, world, coast_amd)
        if rank == 0:
            out["extra"] = legs
    if rank == 0:
        out["full_record"] = write_full_record(out)
        print(final_line(out))
        sys.stdout.flush()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
