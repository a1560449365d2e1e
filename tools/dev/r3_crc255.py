#!/usr/bin/env python3
"""Development probe: the 255-byte crc16 stream (1 GiB) per COAST_CRC_NT value, parity-checked against the byte-serial walk on a sample."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import coast_amd  # noqa: E402
from tools.perf_kernels import timeit  # noqa: E402


def crc_ref(rows):
    out = np.full(rows.shape[0], 0xFFFF, dtype=np.uint32)
    for t in range(rows.shape[1]):
        x = ((out >> 8) ^ rows[:, t]) & 0xFF
        x ^= x >> 4
        out = ((out << 8) ^ (x << 12) ^ (x << 5) ^ x) & 0xFFFF
    return out.astype(np.uint16)


def main():
    eng = coast_amd.Engine(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    nbytes = 1 << 30
    data = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda", generator=g)
    for bl in [int(x) for x in os.environ.get("CRC_BLS", "255").split(",")]:
        nb = nbytes // bl
        out = torch.empty(nb, dtype=torch.int16, device="cuda")
        for rep in (3, 2):
            for nt in os.environ.get("CRC_NTS", "1,2").split(","):
                os.environ["COAST_CRC_NT"] = nt.rstrip("s")
                os.environ["COAST_CRC_WALK"] = "slice4" if nt.endswith("s") else "pair"
                cfg = coast_amd.XmrConfig(rep)
                out.zero_()
                mn, av = timeit(lambda: eng.crc16_batch(data[: nb * bl], bl, out=out, cfg=cfg), reps=5, warm=2)
                idx = np.concatenate([np.arange(2048), np.arange(nb - 2048, nb)])
                rows = data[: nb * bl].view(nb, bl)[torch.from_numpy(idx).cuda()].cpu().numpy().astype(np.uint32)
                got = out[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint16)
                ok = bool((crc_ref(rows) == got).all())
                print("crc16 bl%d rep%d NT=%s %.3f ms %.0f GB/s frac %.3f ok %s" % (bl, rep, nt, mn, nb * bl / mn * 1e-6, nb * bl / mn * 1e-6 / 8000, ok), flush=True)


if __name__ == "__main__":
    main()
