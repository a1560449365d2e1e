import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import coast_amd as ca
from oracle import oracle as orc_mod
from test_gpu_parity import _rand_faults
orc = orc_mod
eng = ca.Engine(0)
rng = np.random.default_rng(int(os.environ.get("SEED", "3")))
batch, n = 3, 256
f = rng.integers(0, 2**32, (batch, n, n), dtype=np.uint32)
s = rng.integers(0, 2**32, (batch, n, n), dtype=np.uint32)
fl = _rand_faults(rng, 120, batch * n * n, 3, [0, 1, 2], n)
dev = lambda a: torch.from_numpy(a.view(np.int32)).cuda()
clean = eng.mm_batch(dev(f), dev(s), cfg=ca.XmrConfig(1)).cpu().numpy().view(np.uint32)
for flags in (False, True):
    det = torch.zeros(batch * n * n, dtype=torch.uint8, device="cuda") if flags else None
    eng.reset_stats()
    eng.inject_faults(fl)
    got = eng.mm_batch(dev(f), dev(s), cfg=ca.XmrConfig(3), detected=det).cpu().numpy().view(np.uint32)
    st = eng.stats()
    d = np.argwhere(got != clean)
    print("flags", flags, "diffs", len(d), "stats", st["errors_corrected"], st["sync_count"], "first", d[:6].tolist())
    items = sorted(int(x["item"]) for x in fl)
    for (b, i, j) in d[:6]:
        it = b * n * n + i * n + j
        print("   diff at", b, i, j, "is fault item:", it in items, "got-clean", (int(got[b, i, j]) - int(clean[b, i, j])) % 2**32)
for row in fl[:0]:
    print(row)
