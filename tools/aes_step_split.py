"""tools/aes_step_split.py -- where the time of an aes-128 DWC step of 1 Mi blocks goes: kernels back to back, + counter fold, + armed upsets,
+ timing events (profiles/r05_aes_step.txt).  Run on the GPU box."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, numpy as np, coast_amd as ca
eng = ca.Engine(0)
n = 1 << 20
g = torch.Generator(device="cuda").manual_seed(1)
st = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda", generator=g)
k = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda", generator=g)
cfg = ca.XmrConfig(ca.DWC)
rng = np.random.default_rng(3)
items = rng.choice(n, 1024, replace=False)
fl = ca.make_faults([(int(it), int(rng.integers(0, 2)), ca.SITE_AES_STATE, int(rng.integers(0, 11)), int(rng.integers(0, 32)), int(rng.integers(0, 4))) for it in items])
def run(name, prof, inject, reduce, steps=400):
    eng.set_profiling(prof)
    d = 0
    def step():
        nonlocal d
        if inject: eng.inject_faults(fl)
        eng.aes128_batch(st, k, d, cfg=cfg); d ^= 1
        if reduce: eng.reduce_counters()
    for _ in range(40): step()
    torch.cuda.synchronize(); eng.reset_stats(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e6
    km = eng.stats()["kernel_ms"] / steps * 1e3 if prof else float("nan")
    print("%-40s step %.1f us  kernel %.1f us" % (name, dt, km), flush=True)
run("clean, no fold, no events", False, False, False)
run("clean, fold, no events", False, False, True)
run("upsets, fold, no events", False, True, True)
run("upsets, fold, events (bench)", True, True, True)
run("upsets, no fold, events", True, True, False)
run("clean, no fold, events", True, False, False)
