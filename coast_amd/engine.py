"""Batch engine: device-resident inputs -> protected kernels -> device-resident outputs + fault counters.

Mirrors the reference's user-facing controls for the hot path: -TMR / -DWC select the replica count
(projects/TMR/TMR.cpp:33, projects/DWC/DWC.cpp:33), -countErrors / -countSyncs expose TMR_ERROR_CNT / __SYNC_COUNT
(projects/dataflowProtection/dataflowProtection.cpp:37,46).  torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib

TMR, DWC, UNPROTECTED = 3, 2, 1
F_NO_STORE_DATA_SYNC = 1  # coast_cfg.flags: the reference's -noStoreDataSync (include/coast_hip.h)
F_BRANCH_SYNC, F_ADDR_SYNC, F_NO_LOAD_SYNC, F_NO_STORE_ADDR_SYNC = 2, 4, 8, 16  # counters inside the SoR (mm, sha256, crc16)
F_LOCAL_STORE_SYNC = 64  # with F_BRANCH_SYNC | F_ADDR_SYNC: the -O0 IR's stores into locals / in-place arrays are data votes (mm, aes128, crc16, cache_test, chsha)
F_O0_SHAPE = 128  # sha256 with F_BRANCH_SYNC | F_ADDR_SYNC: the -O0 IR's shape (padding / output / transform loops are loops with voted counters)
F_CLONE_STAGING = 0x200  # mm side 256 on the matrix cores: the global -> LDS staging loads are cloned and compared -- the DEFAULT since ABI 8 (the flag is accepted)
F_SINGLE_STAGING = 0x400  # ... opt out: every raw word staged once (COAST_F_SINGLE_STAGING)
F_MEMORY_COPIES = 32  # sha256 / aes128 / crc16: arrays hold `replicas` copies back to back; loads per copy, voted stores into every copy


@dataclass(frozen=True)
class XmrConfig:
    replicas: int = TMR   # 3 = -TMR, 2 = -DWC, 1 = no protection
    sync_every: int = 0   # extra loop-condition sync points every V steps (0 = mandatory sync points only)
    flags: int = 0        # F_NO_STORE_DATA_SYNC = the reference's -noStoreDataSync

    def c(self):
        return _lib.CoastCfg(self.replicas, self.sync_every, self.flags)


def make_faults(rows):
    """rows: iterable of (item, replica, site, step, bit[, index]) -> structured array of coast_fault."""
    rows = list(rows)
    f = np.zeros(len(rows), dtype=_lib.FAULT_DTYPE)
    for q, row in enumerate(rows):
        item, replica, site, step, bit = row[:5]
        f[q] = (item, step, replica, site, bit, row[5] if len(row) > 5 else 0)
    return f


def _ptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


class Engine:
    """One context per GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device: int | None = None):
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.CoastLibraryError("no GPU visible: coast_amd only runs on an MI355X (no CPU fallback)")
        self.device = torch.cuda.current_device() if device is None else int(device)
        h = C.c_void_p()
        rc = self._lib.coast_create(C.byref(h), self.device)
        if rc != 0:
            raise RuntimeError("coast_create failed with %d" % rc)
        self._h = h
        # totals live in a torch tensor so that torch.distributed (RCCL) can all-reduce them in place
        self.counters = torch.zeros(4, dtype=torch.int64, device="cuda:%d" % self.device)
        self._check(self._lib.coast_bind_counters(self._h, _ptr(self.counters)))
        self.use_stream(torch.cuda.current_stream(self.device))

    # -- plumbing
    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("libcoast_hip: %s (code %d)" % (self._lib.coast_last_error(self._h).decode(), rc))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.coast_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_stream(self, stream: torch.cuda.Stream):
        self._stream = stream
        self._check(self._lib.coast_set_stream(self._h, C.c_void_p(stream.cuda_stream)))

    # -- fault injector
    def inject_faults(self, faults):
        """Arm single-bit flips (structured array of FAULT_DTYPE) for the next protected launch."""
        f = np.ascontiguousarray(faults, dtype=_lib.FAULT_DTYPE)
        self._check(self._lib.coast_inject_faults(self._h, f.ctypes.data_as(C.c_void_p), len(f)))

    # -- counters
    def reduce_counters(self):
        """Fold the per-workgroup slots into self.counters (async, on the engine's stream)."""
        self._check(self._lib.coast_reduce_counters(self._h))
        return self.counters

    def stats(self) -> dict:
        st = _lib.CoastStats()
        self._check(self._lib.coast_read_stats(self._h, C.byref(st)))
        return {"errors_corrected": int(st.errors_corrected), "sync_count": int(st.sync_count),
                "dwc_detected": int(st.dwc_detected), "launches": int(st.launches),
                "kernel_ms": float(st.kernel_ms), "hbm_bytes": float(st.hbm_bytes)}

    def reset_stats(self):
        self._check(self._lib.coast_reset_stats(self._h))

    def set_profiling(self, on: bool = True):
        """HIP timing events around every protected launch on the engine's stream -> stats()['kernel_ms']."""
        self._check(self._lib.coast_set_profiling(self._h, int(on)))  # True / 1: every launch; n > 1: every n-th launch, counted n times

    def last_launch(self) -> dict:
        """What the most recent protected launch dispatched to (coast_last_launch_info)."""
        li = _lib.CoastLaunchInfo()
        self._check(self._lib.coast_last_launch_info(self._h, C.byref(li)))
        return {"engine": _lib.ENGINE_NAMES.get(int(li.engine), str(li.engine)), "general_blocks": int(li.general_blocks), "hooked_blocks": int(li.hooked_blocks),
                "fast_blocks": int(li.fast_blocks), "armed_faults": int(li.armed_faults),
                "algorithmic_bytes": float(li.algorithmic_bytes)}

    # -- protected regions
    def mm_batch(self, f, s, out=None, cfg: XmrConfig = XmrConfig(), detected=None):
        """f, s: (batch, n, n) int32/uint32-bit-pattern tensors on the GPU.  Returns r with the same layout."""
        assert f.is_cuda and s.is_cuda and f.is_contiguous() and s.is_contiguous()
        assert f.dtype in (torch.int32, torch.uint32) and f.shape == s.shape and f.dim() == 3
        batch, n, _ = f.shape
        if out is None:
            out = torch.empty_like(f)
        cc = cfg.c()
        self._check(self._lib.coast_mm_batch(self._h, _ptr(f), _ptr(s), _ptr(out), n, batch, C.byref(cc),
                                             _ptr(detected) if detected is not None else None))
        return out

    def sha256_batch(self, msgs, length, out=None, cfg: XmrConfig = XmrConfig(), detected=None):
        """msgs: (n_msgs, stride) uint8 on the GPU; message m = first `length` bytes of row m.  With F_MEMORY_COPIES in cfg.flags:
        (replicas, n_msgs, stride) -- replica r hashes copy r, the voted digests come back as (replicas, n_msgs, 32)."""
        copies = bool(cfg.flags & F_MEMORY_COPIES)
        assert msgs.is_cuda and msgs.dtype == torch.uint8 and msgs.dim() == (3 if copies else 2) and msgs.is_contiguous()
        assert not copies or msgs.shape[0] == cfg.replicas
        n, stride = msgs.shape[-2:]
        if out is None:
            out = torch.empty((cfg.replicas, n, 32) if copies else (n, 32), dtype=torch.uint8, device=msgs.device)
        cc = cfg.c()
        self._check(self._lib.coast_sha256_batch(self._h, _ptr(msgs), stride, length, n, _ptr(out), C.byref(cc),
                                                 _ptr(detected) if detected is not None else None))
        return out

    def aes128_batch(self, states, keys, direction, cfg: XmrConfig = XmrConfig(DWC), detected=None):
        """states, keys: (n, 16) uint8 on the GPU, both updated IN PLACE (reference contract); with F_MEMORY_COPIES in cfg.flags
        both are (replicas, n, 16): replica r works on copy r, the voted state / key is stored into every copy."""
        assert states.is_cuda and keys.is_cuda and states.dtype == torch.uint8 and keys.dtype == torch.uint8
        assert states.shape == keys.shape and states.shape[-1] == 16 and states.is_contiguous() and keys.is_contiguous()
        assert states.dim() == (3 if cfg.flags & F_MEMORY_COPIES else 2)
        cc = cfg.c()
        self._check(self._lib.coast_aes128_batch(self._h, _ptr(states), _ptr(keys), states.shape[-2], int(direction),
                                                 C.byref(cc), _ptr(detected) if detected is not None else None))
        return states, keys

    def crc16_batch(self, data, block_len, out=None, cfg: XmrConfig = XmrConfig(), detected=None):
        """data: uint8 tensor holding n_blocks * block_len bytes.  Returns (n_blocks,) crcs as int16 bit patterns.  With
        F_MEMORY_COPIES in cfg.flags: (replicas, n_blocks * block_len) -- replica r walks copy r -- and (replicas, n_blocks) crcs."""
        assert data.is_cuda and data.dtype == torch.uint8 and data.is_contiguous()
        copies = bool(cfg.flags & F_MEMORY_COPIES)
        total = data.numel() // (cfg.replicas if copies else 1)
        nb = total // block_len if block_len else data.shape[0]
        if out is None:
            out = torch.empty((cfg.replicas, nb) if copies else nb, dtype=torch.int16, device=data.device)
        cc = cfg.c()
        self._check(self._lib.coast_crc16_batch(self._h, _ptr(data), block_len, nb, _ptr(out), C.byref(cc),
                                                _ptr(detected) if detected is not None else None))
        return out

    def chsha_batch(self, msgs, length, out=None, cfg: XmrConfig = XmrConfig(), detected=None):
        """CHStone sha (tests/chstone/sha/sha.c): msgs (n_msgs, stride) uint8 on the GPU, `length` a multiple of 64.
        Returns (n_msgs, 5) digests as int32 bit patterns."""
        assert msgs.is_cuda and msgs.dtype == torch.uint8 and msgs.dim() == 2 and msgs.is_contiguous()
        n, stride = msgs.shape
        if out is None:
            out = torch.empty((n, 5), dtype=torch.int32, device=msgs.device)
        cc = cfg.c()
        self._check(self._lib.coast_chsha_batch(self._h, _ptr(msgs), stride, length, n, _ptr(out), C.byref(cc),
                                                _ptr(detected) if detected is not None else None))
        return out

    def cache_test_batch(self, arrays, cfg: XmrConfig = XmrConfig(), detected=None):
        """arrays: (n_arrays, n) int32 on the GPU, scrubbed IN PLACE (calc_sum, tests/cache_test/cacheTest.c:101-177).
        Returns (sums int32, error counts as int32 bit patterns)."""
        assert arrays.is_cuda and arrays.dtype == torch.int32 and arrays.dim() == 2 and arrays.is_contiguous()
        na, n = arrays.shape
        sums = torch.empty(na, dtype=torch.int32, device=arrays.device)
        nerrs = torch.empty(na, dtype=torch.int32, device=arrays.device)
        cc = cfg.c()
        self._check(self._lib.coast_cache_test_batch(self._h, _ptr(arrays), n, na, _ptr(sums), _ptr(nerrs), C.byref(cc),
                                                     _ptr(detected) if detected is not None else None))
        return sums, nerrs

    def quicksort_batch(self, arrays, cfg: XmrConfig = XmrConfig(), detected=None, status=None):
        """arrays: (n_arrays, n) int32 on the GPU, sorted IN PLACE (quick_sort, tests/quicksort/quicksort.c:109-129).
        status (optional uint8 per array): 0 sorted, 1 watchdog, 2 stack overflow."""
        assert arrays.is_cuda and arrays.dtype == torch.int32 and arrays.dim() == 2 and arrays.is_contiguous()
        na, n = arrays.shape
        cc = cfg.c()
        self._check(self._lib.coast_quicksort_batch(self._h, _ptr(arrays), n, na, C.byref(cc),
                                                    _ptr(detected) if detected is not None else None,
                                                    _ptr(status) if status is not None else None))
        return arrays

    def chaes_batch(self, states, keys, type_, dir_=0, cfg: XmrConfig = XmrConfig(), detected=None):
        """CHStone aes (tests/chstone/aes): Rijndael, type_ = key bits * 1000 + block bits.  states (n, 4 Nb) uint8 IN PLACE,
        keys (n, 4 Nk) uint8 untouched."""
        assert states.is_cuda and keys.is_cuda and states.dtype == torch.uint8 and keys.dtype == torch.uint8
        assert states.is_contiguous() and keys.is_contiguous() and states.dim() == 2 and keys.dim() == 2
        nk, nb = type_ // 1000 // 32, type_ % 1000 // 32
        assert states.shape[1] == 4 * nb and keys.shape[1] == 4 * nk and states.shape[0] == keys.shape[0]
        cc = cfg.c()
        self._check(self._lib.coast_chaes_batch(self._h, _ptr(states), _ptr(keys), states.shape[0], int(type_), int(dir_),
                                                C.byref(cc), _ptr(detected) if detected is not None else None))
        return states

    def crazycf_batch(self, params, cfcss=True, results=None, status=None):
        """params: (n, 3) int32 on the GPU, rows of (seed, size, timesThroughWhile): n runs of tests/crazyCF/crazyCF.c's main()
        under control-flow signatures (projects/CFCSS; cfcss=False runs them bare).  Returns (results (n, 4) int32 =
        total, printed, n_prints, blocks; status uint8 = 0 ok, 1 FAULT_DETECTED_CFC, 2 watchdog, 3 jumped out of the program)."""
        assert params.is_cuda and params.dtype == torch.int32 and params.dim() == 2 and params.shape[1] == 3
        assert params.is_contiguous()
        n = params.shape[0]
        if results is None:
            results = torch.zeros((n, 4), dtype=torch.int32, device=params.device)
        if status is None:
            status = torch.zeros(n, dtype=torch.uint8, device=params.device)
        self._check(self._lib.coast_crazycf_batch(self._h, _ptr(params), n, _ptr(results), _ptr(status), int(bool(cfcss))))
        return results, status

    def crazycf_xmr_batch(self, params, cfg: XmrConfig = XmrConfig(), results=None, status=None, detected=None):
        """params: (n, 3) int32 rows of (seed, size, timesThroughWhile): n runs of tests/crazyCF/crazyCF.c's main() under -TMR / -DWC
        (unittest/cfg/full_tmr.yml:8), the printf arguments voted; F_BRANCH_SYNC / F_ADDR_SYNC / F_LOCAL_STORE_SYNC add the loop
        conditions, the switch operand, array[i]'s offset, the stored data.  Returns (results (n, 4) int32, status uint8)."""
        assert params.is_cuda and params.dtype == torch.int32 and params.dim() == 2 and params.shape[1] == 3
        assert params.is_contiguous()
        n = params.shape[0]
        if results is None:
            results = torch.zeros((n, 4), dtype=torch.int32, device=params.device)
        if status is None:
            status = torch.zeros(n, dtype=torch.uint8, device=params.device)
        cc = cfg.c()
        self._check(self._lib.coast_crazycf_xmr_batch(self._h, _ptr(params), n, _ptr(results), _ptr(status), C.byref(cc),
                                                      _ptr(detected) if detected is not None else None))
        return results, status

    # -- default mode (memory replicated x3 / x2): vote where the copies re-converge
    def sync_copies(self, copies, out=None, scrub=True, detected=None, fp=False, vector_width=1):
        """copies: 3 (TMR) or 2 (DWC) equally shaped contiguous GPU tensors holding the per-copy results of replicas=1
        launches.  Votes / compares them word by word (32-bit), counts into the engine's counters, optionally repairs
        the copies in place, and returns the voted tensor.  fp / vector_width: the pass's operand-type rules (float words compared
        with fcmp oeq; IR vectors counted per lane without a __SYNC_COUNT increment; coast_sync_copies_typed)."""
        assert len(copies) in (2, 3)
        t0 = copies[0]
        nbytes = t0.numel() * t0.element_size()
        assert all(t.is_cuda and t.is_contiguous() and t.numel() * t.element_size() == nbytes for t in copies)
        if out is None:
            out = torch.empty_like(t0)
        arr = (C.c_void_p * len(copies))(*[t.data_ptr() for t in copies])
        self._check(self._lib.coast_sync_copies_typed(self._h, arr, len(copies), nbytes, _ptr(out), int(bool(scrub)),
                                                      _ptr(detected) if detected is not None else None, 1 if fp else 0,
                                                      int(vector_width)))
        return out

    def flip_memory(self, tensor, byte_offset, bit):
        """injectFaultMem analogue: flip one bit of device memory (ordered on the engine's stream)."""
        self._check(self._lib.coast_flip_memory(self._h, _ptr(tensor), int(byte_offset), int(bit)))
