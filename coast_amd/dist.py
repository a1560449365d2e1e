"""Multi-GPU: one process per GPU, independent protected blocks sharded across ranks, and ONE collective -- the
all-reduce of the fault counters over RCCL/xGMI.  It replaces the reference's single global `TMR_ERROR_CNT += 1`
(projects/dataflowProtection/synchronization.cpp:1428-1431); the data never moves between GPUs.

The message is 4 x int64 (32 bytes): pure latency on xGMI, so it is issued once per batch, after the local
per-workgroup slots were folded on the device (Engine.reduce_counters)."""
from __future__ import annotations

import torch


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block partition: rank r owns [lo, hi).  The first (n_items % world) ranks get one extra item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_counters(engine, dist=None, group=None, snapshot: bool = True) -> torch.Tensor:
    """Global {errors_corrected, sync_count, dwc_detected, launches}.  Local totals stay untouched (cumulative).
    snapshot=False with no process group: the live totals tensor itself (no copy kernel; it keeps counting)."""
    if not snapshot and (dist is None or not dist.is_initialized()):
        return engine.counters
    tot = engine.counters.clone()
    if dist is not None and dist.is_initialized():  # also at world size 1: the same RCCL call the N-rank job issues
        if tot.is_cuda and dist.get_backend(group) == "gloo":  # dry runs of the rank logic without RCCL: stage through the host
            host = tot.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            tot = host.to(tot.device)
        else:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    return tot


def any_dwc_detected(engine, dist=None, group=None) -> bool:
    """DWC abort semantics across the job: any GPU with dwc_detected > 0 means FAULT_DETECTED_DWC() would have fired."""
    return int(allreduce_counters(engine, dist, group)[2].item()) > 0
