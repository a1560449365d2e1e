"""ctypes binding of include/coast_hip.h.  Fails loudly when the HIP library is missing: there is no CPU path."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

FAULT_DTYPE = np.dtype(
    [("item", "<u8"), ("step", "<u4"), ("replica", "u1"), ("site", "u1"), ("bit", "u1"), ("index", "u1")]
)
assert FAULT_DTYPE.itemsize == 16


class CoastCfg(C.Structure):
    _fields_ = [("replicas", C.c_uint32), ("sync_every", C.c_uint32), ("flags", C.c_uint32)]


class CoastStats(C.Structure):
    _fields_ = [("errors_corrected", C.c_uint64), ("sync_count", C.c_uint64), ("dwc_detected", C.c_uint64),
                ("launches", C.c_uint64), ("kernel_ms", C.c_double), ("hbm_bytes", C.c_double)]


class CoastLaunchInfo(C.Structure):
    _fields_ = [("engine", C.c_uint32), ("hooked_blocks", C.c_uint32), ("general_blocks", C.c_uint64),
                ("fast_blocks", C.c_uint64), ("armed_faults", C.c_uint64), ("algorithmic_bytes", C.c_double)]


class CoastCfcGraph(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("flags", C.POINTER(C.c_uint8)), ("func", C.POINTER(C.c_uint16)),
                ("succ_begin", C.POINTER(C.c_uint32)), ("succ", C.POINTER(C.c_uint16)), ("n_calls", C.c_uint32),
                ("call_node", C.POINTER(C.c_uint16)), ("call_entry", C.POINTER(C.c_uint16)), ("main_func", C.c_uint32)]


class CoastCfcTables(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_buffers", C.c_uint32), ("sig", C.c_uint16 * 256), ("sig_diff", C.c_uint16 * 256),
                ("sig_adj", C.c_uint16 * 256), ("flags", C.c_uint8 * 256), ("succ_begin", C.c_uint32 * 257),
                ("succ", C.c_uint16 * 1024), ("call_pre_adj", C.c_uint16 * 64), ("call_post_adj", C.c_uint16 * 64)]


CRAZYCF_PARAMS_DTYPE = np.dtype([("seed", "<i4"), ("size", "<i4"), ("times", "<i4")])
CRAZYCF_RESULT_DTYPE = np.dtype([("total", "<i4"), ("printed", "<i4"), ("n_prints", "<u4"), ("blocks", "<u4")])

ENGINE_NAMES = {0: "none", 1: "valu", 2: "matrix_core", 3: "stepwise", 4: "vote"}


# every symbol include/coast_hip.h declares (tests check the library exports all of them)
SYMBOLS = {
    "coast_abi_version": (C.c_int, []),
    "coast_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "coast_destroy": (None, [C.c_void_p]),
    "coast_last_error": (C.c_char_p, [C.c_void_p]),
    "coast_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "coast_bind_counters": (C.c_int, [C.c_void_p, C.c_void_p]),
    "coast_reduce_counters": (C.c_int, [C.c_void_p]),
    "coast_allreduce_counters": (C.c_int, [C.c_void_p, C.c_void_p]),
    "coast_read_stats": (C.c_int, [C.c_void_p, C.POINTER(CoastStats)]),
    "coast_reset_stats": (C.c_int, [C.c_void_p]),
    "coast_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "coast_last_launch_info": (C.c_int, [C.c_void_p, C.POINTER(CoastLaunchInfo)]),
    "coast_source_hash": (C.c_char_p, []),
    "coast_inject_faults": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "coast_mm_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t,
                                 C.POINTER(CoastCfg), C.c_void_p]),
    "coast_sha256_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_void_p,
                                     C.POINTER(CoastCfg), C.c_void_p]),
    "coast_aes128_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(CoastCfg),
                                     C.c_void_p]),
    "coast_crc16_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p, C.POINTER(CoastCfg),
                                    C.c_void_p]),
    "coast_cache_test_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.POINTER(CoastCfg), C.c_void_p]),
    "coast_chsha_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_void_p,
                                    C.POINTER(CoastCfg), C.c_void_p]),
    "coast_chaes_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(CoastCfg),
                                    C.c_void_p]),
    "coast_quicksort_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.POINTER(CoastCfg), C.c_void_p,
                                        C.c_void_p]),
    "coast_cfcss_assign": (C.c_int, [C.POINTER(CoastCfcGraph), C.POINTER(CoastCfcTables)]),
    "coast_crazycf_graph": (C.c_int, [C.POINTER(CoastCfcGraph)]),
    "coast_crazycf_tables": (C.c_int, [C.POINTER(CoastCfcTables)]),
    "coast_crazycf_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]),
    "coast_crazycf_xmr_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(CoastCfg), C.c_void_p]),
    "coast_sync_copies": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_void_p, C.c_int,
                                    C.c_void_p]),
    "coast_sync_copies_typed": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_int, C.c_uint32]),
    "coast_flip_memory": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint]),
    "coast_matrix_multiply_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(CoastCfg)]),
    "coast_sha256_host": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(CoastCfg)]),
    "coast_aes_enc_dec_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint8, C.POINTER(CoastCfg)]),
    "coast_crc16_host": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(CoastCfg)]),
    "coast_cache_test_host": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(CoastCfg)]),
    "coast_chsha_host": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(CoastCfg)]),
    "coast_quicksort_host": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(CoastCfg)]),
    "coast_chaes_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(CoastCfg)]),
    "coast_host_inject_faults": (C.c_int, [C.c_void_p, C.c_size_t]),
    "coast_host_stats": (C.c_int, [C.POINTER(CoastStats), C.c_int]),
}

_lib = None


class CoastLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    return _build.LIB


def load():
    """dlopen libcoast_hip.so (built in-tree).  Raises CoastLibraryError when it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    override = os.environ.get("COAST_LIB_OVERRIDE")  # development: A/B a differently built library in one GPU session
    if override:
        L = C.CDLL(override)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
        return L
    path = lib_path()
    # A fresh checkout or edited sources: (re)compile the HIP library in-tree -- still the native path, there is nothing
    # to fall back to.  build() is a no-op when the library was built from exactly these sources (content hash, not
    # mtimes: the prebuilt .so travels to the GPU box with the snapshot).
    try:
        _build.build()
    except Exception as e:  # no hipcc, compile error ...
        raise CoastLibraryError(
            "%s is missing or stale and could not be built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "coast_amd has no CPU fallback." % (path, e)) from e
    try:
        L = C.CDLL(path)
    except OSError as e:
        raise CoastLibraryError("cannot load %s: %s" % (path, e)) from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise CoastLibraryError("%s does not export %s" % (path, name)) from e
        fn.restype = res
        fn.argtypes = args
    want = _build.source_hash()
    got = L.coast_source_hash().decode()
    if got != want:
        raise CoastLibraryError("%s was built from other sources (library %s, tree %s): rebuild it" % (path, got, want))
    _lib = L
    return L
