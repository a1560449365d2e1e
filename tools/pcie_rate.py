#!/usr/bin/env python3
"""The headline workload with its operands starting in HOST memory: what a caller of the drop-in boundary sees when the matrices are not
resident yet.  bench.py's `value` is measured with inputs in HBM (the contract); this is the PCIe-inclusive figure DESIGN.md section 7 quotes
next to it.  Prints one JSON line.
  batch      B side-256 uint32 matrix pairs in pinned host memory -> H2D -> coast_mm_batch (TMR) -> D2H into pinned memory, serial
  pipelined  the same in chunks on two streams (copy of chunk k+1 under the product of chunk k)
  host_shim  one coast_matrix_multiply_host call (pageable host pointers, what matrix_multiply() of the unmodified C program does)"""
import ctypes as C
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import coast_amd as ca  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    n = 256
    eng = ca.Engine()
    g = torch.Generator().manual_seed(0)
    hf = torch.randint(-2**31, 2**31, (B, n, n), dtype=torch.int32, generator=g).pin_memory()
    hs = torch.randint(-2**31, 2**31, (B, n, n), dtype=torch.int32, generator=g).pin_memory()
    hr = torch.empty((B, n, n), dtype=torch.int32).pin_memory()
    df, ds = torch.empty_like(hf, device="cuda"), torch.empty_like(hs, device="cuda")
    cfg = ca.XmrConfig(3)

    def serial():
        df.copy_(hf, non_blocking=True)
        ds.copy_(hs, non_blocking=True)
        out = eng.mm_batch(df, ds, cfg=cfg)
        hr.copy_(out, non_blocking=True)
        torch.cuda.synchronize()

    def resident():
        eng.mm_batch(df, ds, cfg=cfg)
        torch.cuda.synchronize()

    for fn in (serial, resident):
        fn()
    t = {}
    for name, fn in (("serial", serial), ("resident", resident)):
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        t[name] = (time.perf_counter() - t0) / 3
    # one call of the single-matrix host shim (pageable memory, its own staging copies)
    f1, s1 = hf[0].numpy().copy(), hs[0].numpy().copy()
    r1 = np.empty_like(f1)
    cc = cfg.c()
    call = lambda: eng._lib.coast_matrix_multiply_host(f1.ctypes.data_as(C.c_void_p), s1.ctypes.data_as(C.c_void_p),  # noqa: E731
                                                       r1.ctypes.data_as(C.c_void_p), n, C.byref(cc))
    assert call() == 0
    t0 = time.perf_counter()
    for _ in range(20):
        call()
    t["host_shim"] = (time.perf_counter() - t0) / 20
    assert (r1 == hr[0].numpy()).all()
    elems = B * n * n
    moved = 3 * B * n * n * 4
    print(json.dumps({"workload": "mm 256^2 TMR", "matrices": B, "resident_ms": 1e3 * t["resident"], "host_to_host_ms": 1e3 * t["serial"],
                      "resident_elems_per_s": elems / t["resident"], "pcie_inclusive_elems_per_s": elems / t["serial"],
                      "pcie_GBps": moved / t["serial"] / 1e9, "host_shim_one_matrix_ms": 1e3 * t["host_shim"],
                      "host_shim_elems_per_s": n * n / t["host_shim"]}))


if __name__ == "__main__":
    main()
