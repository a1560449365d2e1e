"""ctypes front-end to the CPU oracle (oracle/liboracle.so) and, when present, to the reference's own C
kernels compiled unmodified (oracle/_ref/libcoast_ref.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never from coast_amd/ (the product fails loudly without its HIP library instead of falling back to this).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

SITE_MM_ACC, SITE_MM_OPA, SITE_MM_OPB = 0, 1, 2
SITE_SHA_M, SITE_SHA_WV, SITE_SHA_STATE = 8, 9, 10
SITE_AES_STATE, SITE_AES_KEY = 16, 17
SITE_CRC_CRC, SITE_CRC_X = 24, 25

# identical layout to orc_fault / coast_fault (16 bytes)
FAULT_DTYPE = np.dtype(
    [("item", "<u8"), ("step", "<u4"), ("replica", "u1"), ("site", "u1"), ("bit", "u1"), ("index", "u1")]
)
assert FAULT_DTYPE.itemsize == 16


class Cfg(C.Structure):
    _fields_ = [("replicas", C.c_uint32), ("sync_every", C.c_uint32), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [
        ("errors_corrected", C.c_uint64),
        ("sync_count", C.c_uint64),
        ("dwc_detected", C.c_uint64),
        ("reserved", C.c_uint64),
    ]

    def as_dict(self):
        return {
            "errors_corrected": int(self.errors_corrected),
            "sync_count": int(self.sync_count),
            "dwc_detected": int(self.dwc_detected),
        }


def build(force: bool = False) -> None:
    """make -C oracle (compiles liboracle.so; also oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
        for f in ("coast_oracle.c", "coast_oracle.h", "cpu_tmr_baseline.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/tests"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.orc_mm_xor.restype = C.c_uint32
        L.orc_crc16_plain.restype = C.c_uint16
        L.orc_cpu_tmr_mm.restype = C.c_int
        L.orc_cpu_tmr_mm_threads.restype = C.c_double
        L.orc_cpu_tmr_mm_campaign.restype = C.c_uint64
        L.orc_aes_sbox.restype = C.POINTER(C.c_uint8)
        L.orc_aes_rsbox.restype = C.POINTER(C.c_uint8)
        _lib = L
    return _lib


def ref():
    """The reference's own kernels (None when oracle/_ref was never built)."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libcoast_ref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_crc16.restype = C.c_uint16
        R.ref_crc16.argtypes = [C.c_char_p, C.c_ubyte]
        _ref = R
    return _ref


def ref_chsha(data):
    """the reference's CHStone sha (sha_init / sha_update / sha_final) over `data` (len % 64 == 0) -> 5 uint32"""
    d = (C.c_uint32 * 5)()
    buf = bytes(data)
    ref().ref_chsha(buf, C.c_int(len(buf)), d)
    return np.array(list(d), dtype=np.uint32)


def ref_chsha_vectors():
    """the benchmark's built-in input (2 x 8192 bytes, sha_data.c) and the digest of the reference's own sha_stream()"""
    R = ref()
    arr = (C.c_ubyte * (2 * 8192)).in_dll(R, "ref_chsha_indata")
    d = (C.c_uint32 * 5)()
    R.ref_chsha_stream(d)
    return np.frombuffer(bytes(arr), np.uint8).reshape(2, 8192).copy(), np.array(list(d), dtype=np.uint32)


REPLICA_ALL = 255  # include/coast_hip.h COAST_REPLICA_ALL
F_LOCAL_STORE_SYNC = 64  # ORC_F_LOCAL_STORE_SYNC: with BRANCH_SYNC | ADDR_SYNC, the data votes of the -O0 IR's stores into locals / in-place arrays
F_O0_SHAPE = 128  # ORC_F_O0_SHAPE: sha256 with BRANCH_SYNC | ADDR_SYNC: the -O0 IR's shape (padding / output / transform loops are loops)
F_MEMORY_COPIES = 32  # ORC_F_MEMORY_COPIES: arrays are (replicas, n, ...) -- replica r works on copy r, stores are voted into every copy


def _faults(faults):
    """A common-mode upset (replica == REPLICA_ALL: state the replicas share) is the same flip in every replica's copy of the
    value: stated here as one fault per replica -- the C model ignores replica numbers the run does not have."""
    if faults is None:
        return np.zeros(0, dtype=FAULT_DTYPE)
    f = np.ascontiguousarray(faults, dtype=FAULT_DTYPE)
    common = f["replica"] == REPLICA_ALL
    if common.any():
        parts = [f[~common]]
        for r in range(3):
            c = f[common].copy()
            c["replica"] = r
            parts.append(c)
        f = np.ascontiguousarray(np.concatenate(parts))
    return f


def make_faults(rows):
    """rows: iterable of (item, replica, site, step, bit[, index])"""
    rows = list(rows)
    f = np.zeros(len(rows), dtype=FAULT_DTYPE)
    for q, row in enumerate(rows):
        item, replica, site, step, bit = row[:5]
        f[q] = (item, step, replica, site, bit, row[5] if len(row) > 5 else 0)
    return f


# ---------------------------------------------------------------- plain kernels
def mm_plain(f, s):
    f = np.ascontiguousarray(f, dtype=np.uint32)
    s = np.ascontiguousarray(s, dtype=np.uint32)
    n = f.shape[-1]
    r = np.empty((n, n), dtype=np.uint32)
    lib().orc_mm_plain(_p(f, C.c_uint32), _p(s, C.c_uint32), _p(r, C.c_uint32), C.c_int(n))
    return r


def mm_xor(r):
    r = np.ascontiguousarray(r, dtype=np.uint32)
    return int(lib().orc_mm_xor(_p(r, C.c_uint32), C.c_int(r.shape[-1])))


def sha256_plain(data: bytes) -> bytes:
    buf = np.frombuffer(bytes(data) + b"\0", dtype=np.uint8).copy()
    out = np.empty(32, dtype=np.uint8)
    lib().orc_sha256_plain(_p(buf, C.c_uint8), C.c_uint32(len(data)), _p(out, C.c_uint8))
    return out.tobytes()


def aes128_plain(state: bytes, key: bytes, direction: int):
    s = np.frombuffer(bytes(state), dtype=np.uint8).copy()
    k = np.frombuffer(bytes(key), dtype=np.uint8).copy()
    lib().orc_aes128_plain(_p(s, C.c_uint8), _p(k, C.c_uint8), C.c_uint8(direction))
    return s.tobytes(), k.tobytes()


def crc16_plain(data: bytes) -> int:
    buf = np.frombuffer(bytes(data) + b"\0", dtype=np.uint8).copy()
    return int(lib().orc_crc16_plain(_p(buf, C.c_uint8), C.c_uint32(len(data))))


# ---------------------------------------------------------------- replicated model
def mm_xmr(f, s, replicas=3, sync_every=0, faults=None, flags=0):
    """f, s: (batch, n, n) uint32.  Returns (r, stats dict, detected per item)."""
    f = np.ascontiguousarray(f, dtype=np.uint32)
    s = np.ascontiguousarray(s, dtype=np.uint32)
    if f.ndim == 2:
        f, s = f[None], s[None]
    batch, n, _ = f.shape
    r = np.empty_like(f)
    fl = _faults(faults)
    st = Stats()
    det = np.zeros(batch * n * n, dtype=np.uint8)
    cfg = Cfg(replicas, sync_every, flags)
    lib().orc_mm_xmr(_p(f, C.c_uint32), _p(s, C.c_uint32), _p(r, C.c_uint32), C.c_int(n), C.c_size_t(batch),
                     C.byref(cfg), fl.ctypes.data_as(C.c_void_p), C.c_size_t(len(fl)), C.byref(st),
                     _p(det, C.c_uint8))
    return r, st.as_dict(), det


def mm_xmr_items(f, s, items, replicas=3, sync_every=0, faults=None, flags=0):
    f = np.ascontiguousarray(f, dtype=np.uint32)
    s = np.ascontiguousarray(s, dtype=np.uint32)
    n = f.shape[-1]
    items = np.ascontiguousarray(items, dtype=np.uint64)
    out = np.empty(len(items), dtype=np.uint32)
    fl = _faults(faults)
    st = Stats()
    det = np.zeros(len(items), dtype=np.uint8)
    cfg = Cfg(replicas, sync_every, flags)
    lib().orc_mm_xmr_items(_p(f, C.c_uint32), _p(s, C.c_uint32), C.c_int(n), _p(items, C.c_uint64),
                           C.c_size_t(len(items)), _p(out, C.c_uint32), C.byref(cfg),
                           fl.ctypes.data_as(C.c_void_p), C.c_size_t(len(fl)), C.byref(st), _p(det, C.c_uint8))
    return out, st.as_dict(), det


def sha256_xmr(msgs, length, replicas=3, faults=None, flags=0):
    """msgs: (nmsgs, stride) uint8, each message = first `length` bytes of its row."""
    msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
    copies = bool(flags & F_MEMORY_COPIES)  # msgs: (replicas, nmsgs, stride); the digests come back as (replicas, nmsgs, 32)
    if copies:
        assert msgs.ndim == 3 and msgs.shape[0] == replicas
    nm, stride = msgs.shape[-2:]
    dig = np.empty((replicas, nm, 32) if copies else (nm, 32), dtype=np.uint8)
    fl = _faults(faults)
    st = Stats()
    det = np.zeros(nm, dtype=np.uint8)
    cfg = Cfg(replicas, 0, flags)
    pad = np.concatenate([msgs.reshape(-1), np.zeros(8, np.uint8)])  # keep 0-length rows addressable
    lib().orc_sha256_xmr(_p(pad, C.c_uint8), C.c_size_t(stride), C.c_uint32(length), C.c_size_t(nm),
                         _p(dig, C.c_uint8), C.byref(cfg), fl.ctypes.data_as(C.c_void_p), C.c_size_t(len(fl)),
                         C.byref(st), _p(det, C.c_uint8))
    return dig, st.as_dict(), det


def aes128_xmr(states, keys, direction, replicas=2, sync_every=0, faults=None, flags=0):
    """states, keys: (n, 16) uint8; returns new (states, keys, stats, detected) -- inputs untouched."""
    s = np.array(states, dtype=np.uint8, copy=True, order="C")
    k = np.array(keys, dtype=np.uint8, copy=True, order="C")
    if flags & F_MEMORY_COPIES:  # (replicas, n, 16) each
        assert s.ndim == 3 and s.shape[0] == replicas and k.shape == s.shape
    n = s.shape[-2]
    fl = _faults(faults)
    st = Stats()
    det = np.zeros(n, dtype=np.uint8)
    cfg = Cfg(replicas, sync_every, flags)
    lib().orc_aes128_xmr(_p(s, C.c_uint8), _p(k, C.c_uint8), C.c_size_t(n), C.c_int(direction), C.byref(cfg),
                         fl.ctypes.data_as(C.c_void_p), C.c_size_t(len(fl)), C.byref(st), _p(det, C.c_uint8))
    return s, k, st.as_dict(), det


def crc16_xmr(data, block_len, replicas=3, sync_every=0, faults=None, flags=0):
    """data: (nblocks, block_len) uint8."""
    copies = bool(flags & F_MEMORY_COPIES)  # data: (replicas, nblocks, block_len); crcs come back as (replicas, nblocks)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    if copies:
        assert data.ndim == 3 and data.shape[0] == replicas and data.shape[2] == block_len
        nb = data.shape[1]
    else:
        data = np.ascontiguousarray(data.reshape(-1, max(block_len, 1))[:, :block_len])
        nb = data.shape[0]
    crcs = np.empty((replicas, nb) if copies else nb, dtype=np.uint16)
    fl = _faults(faults)
    st = Stats()
    det = np.zeros(nb, dtype=np.uint8)
    cfg = Cfg(replicas, sync_every, flags)
    pad = np.concatenate([data.reshape(-1), np.zeros(8, np.uint8)])
    lib().orc_crc16_xmr(_p(pad, C.c_uint8), C.c_uint32(block_len), C.c_size_t(nb), _p(crcs, C.c_uint16),
                        C.byref(cfg), fl.ctypes.data_as(C.c_void_p), C.c_size_t(len(fl)), C.byref(st),
                        _p(det, C.c_uint8))
    return crcs, st.as_dict(), det


def chsha_xmr(msgs, length, replicas=3, faults=None, flags=0):
    """CHStone sha (tests/chstone/sha/sha.c): msgs (nmsgs, stride) uint8, `length` a multiple of 64; returns
    (digests (nmsgs, 5) uint32, stats, detected)."""
    msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
    nm, stride = msgs.shape
    assert length % 64 == 0 and length <= stride
    dig = np.empty((nm, 5), dtype=np.uint32)
    fl = _faults(faults)
    st = Stats()
    det = np.zeros(nm, dtype=np.uint8)
    cfg = Cfg(replicas, 0, flags)
    pad = np.concatenate([msgs.reshape(-1), np.zeros(8, np.uint8)])
    lib().orc_chsha_xmr(_p(pad, C.c_uint8), C.c_size_t(stride), C.c_uint32(length), C.c_size_t(nm), _p(dig, C.c_uint32),
                        C.byref(cfg), fl.ctypes.data_as(C.c_void_p), C.c_size_t(len(fl)), C.byref(st), _p(det, C.c_uint8))
    return dig, st.as_dict(), det


def cache_test_xmr(arrays, replicas=3, faults=None, flags=0):
    """arrays: (narrays, n) int32; returns (scrubbed arrays, sums, error counts, stats, detected) -- calc_sum of
    tests/cache_test/cacheTest.c:101-177 under the protection model."""
    a = np.array(np.ascontiguousarray(arrays, dtype=np.int32), copy=True)
    na, n = a.shape
    sums = np.empty(na, dtype=np.int32)
    nerrs = np.empty(na, dtype=np.uint32)
    fl = _faults(faults)
    st = Stats()
    det = np.zeros(na, dtype=np.uint8)
    cfg = Cfg(replicas, 0, flags)
    lib().orc_cache_test_xmr(_p(a, C.c_int32), C.c_uint32(n), C.c_size_t(na), _p(sums, C.c_int32), _p(nerrs, C.c_uint32),
                             C.byref(cfg), fl.ctypes.data_as(C.c_void_p), C.c_size_t(len(fl)), C.byref(st),
                             _p(det, C.c_uint8))
    return a, sums, nerrs, st.as_dict(), det


def sync_copies(copies, scrub=True, fp=False, vector_width=1):
    """Default-mode exit vote: copies = 3 (TMR) or 2 (DWC) arrays of 32-bit words.  fp: the words are floats (fcmp oeq / one);
    vector_width > 1: IR vectors of that many lanes (per-lane count, no __SYNC_COUNT increment).  Returns (voted, copies after
    scrub, stats, detected per word)."""
    cs = [np.array(np.ascontiguousarray(c).view(np.uint32).reshape(-1), copy=True) for c in copies]
    n = cs[0].size
    voted = np.empty(n, dtype=np.uint32)
    det = np.zeros(n, dtype=np.uint8)
    st = Stats()
    c2 = cs[2] if len(cs) == 3 else cs[0]
    lib().orc_sync_copies_typed(_p(cs[0], C.c_uint32), _p(cs[1], C.c_uint32), _p(c2, C.c_uint32), C.c_int(len(cs)),
                                C.c_size_t(n), _p(voted, C.c_uint32), C.c_int(int(scrub)), C.byref(st), _p(det, C.c_uint8),
                                C.c_int(int(fp)), C.c_uint32(vector_width))
    return voted, cs, st.as_dict(), det


def quicksort_plain(arr):
    a = np.ascontiguousarray(arr, dtype=np.int32).copy()
    lib().orc_quicksort_plain(_p(a, C.c_int32), C.c_uint32(a.shape[-1]))
    return a


def quicksort_xmr(arrays, replicas=3, faults=None, flags=0):
    """arrays: (narrays, n) int32.  Returns (sorted copies, stats dict, detected per array, status per array)."""
    a = np.ascontiguousarray(arrays, dtype=np.int32).copy()
    na, n = a.shape
    fl = _faults(faults)
    st = Stats()
    det = np.zeros(na, dtype=np.uint8)
    status = np.zeros(na, dtype=np.uint8)
    cfg = Cfg(replicas, 0, flags)
    lib().orc_quicksort_xmr(_p(a, C.c_int32), C.c_uint32(n), C.c_size_t(na), C.byref(cfg), fl.ctypes.data_as(C.c_void_p),
                            C.c_size_t(len(fl)), C.byref(st), _p(det, C.c_uint8), _p(status, C.c_uint8))
    return a, st.as_dict(), det, status


# ---- CHStone aes (chaes_oracle.inc) ----
CHAES_TYPES = (128128, 128192, 128256, 192128, 192192, 192256, 256128, 256192, 256256)
SITE_CHAES_STATE, SITE_CHAES_WORD = 64, 65


def chaes_geom(type_):
    nk, nb = type_ // 1000 // 32, type_ % 1000 // 32
    return nk, nb, max(nk, nb) + 6


def chaes_xmr(states, keys, type_, dir_=0, replicas=3, sync_every=0, faults=None, flags=0):
    """states: (n, 4 Nb) uint8, keys: (n, 4 Nk) uint8.  Returns (states out, stats dict, detected per block)."""
    nk, nb, _ = chaes_geom(type_)
    s = np.ascontiguousarray(states, dtype=np.uint8).reshape(-1, 4 * nb).copy()
    k = np.ascontiguousarray(keys, dtype=np.uint8).reshape(-1, 4 * nk)
    n = s.shape[0]
    assert k.shape[0] == n
    fl = _faults(faults)
    st = Stats()
    det = np.zeros(n, dtype=np.uint8)
    cfg = Cfg(replicas, sync_every, flags)
    rc = lib().orc_chaes_xmr(_p(s, C.c_uint8), _p(k, C.c_uint8), C.c_size_t(n), C.c_int(type_), C.c_int(dir_), C.byref(cfg),
                             fl.ctypes.data_as(C.c_void_p), C.c_size_t(len(fl)), C.byref(st), _p(det, C.c_uint8))
    if rc:
        raise ValueError("orc_chaes_xmr: unknown type %d" % type_)
    return s, st.as_dict(), det


def chaes_plain(state, key, type_, dir_=0):
    s = np.ascontiguousarray(state, dtype=np.uint8).copy()
    k = np.ascontiguousarray(key, dtype=np.uint8)
    rc = lib().orc_chaes_plain(_p(s, C.c_uint8), _p(k, C.c_uint8), C.c_int(type_), C.c_int(dir_))
    if rc:
        raise ValueError("orc_chaes_plain: unknown type %d" % type_)
    return s


def ref_chaes(state, key, type_, dir_=0):
    """the reference's own encrypt / decrypt (tests/chstone/aes, compiled from where it lies): one byte per int"""
    s = np.ascontiguousarray(state, dtype=np.uint8).astype(np.int32)
    s = np.concatenate([s, np.zeros(32 - len(s), np.int32)])
    k = np.ascontiguousarray(key, dtype=np.uint8).astype(np.int32)
    k = np.concatenate([k, np.zeros(32 - len(k), np.int32)])
    rc = ref().ref_chaes(_p(s, C.c_int32), _p(k, C.c_int32), C.c_int(type_), C.c_int(dir_))
    if rc:
        raise ValueError("ref_chaes: type %d" % type_)
    nb = type_ % 1000 // 32
    assert ((s >= 0) & (s < 256)).all()
    return s[:4 * nb].astype(np.uint8)


# ---- CFCSS (cfcss_oracle.c) ----
class CfcGraph(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("flags", C.POINTER(C.c_uint8)), ("func", C.POINTER(C.c_uint16)),
                ("succ_begin", C.POINTER(C.c_uint32)), ("succ", C.POINTER(C.c_uint16)), ("n_calls", C.c_uint32),
                ("call_node", C.POINTER(C.c_uint16)), ("call_entry", C.POINTER(C.c_uint16)), ("main_func", C.c_uint32)]


class CfcTables(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_buffers", C.c_uint32), ("sig", C.c_uint16 * 256), ("sig_diff", C.c_uint16 * 256),
                ("sig_adj", C.c_uint16 * 256), ("flags", C.c_uint8 * 256), ("succ_begin", C.c_uint32 * 257),
                ("succ", C.c_uint16 * 1024), ("call_pre_adj", C.c_uint16 * 64), ("call_post_adj", C.c_uint16 * 64)]


CFC_FAN_IN, CFC_CHECKED, CFC_BUFFER, CFC_SKIP, CFC_RET = 1, 2, 4, 8, 16
CFC_OK, CFC_DETECTED, CFC_WATCHDOG, CFC_WILD = 0, 1, 2, 3
SITE_CFC_PC, SITE_CFC_RTS, SITE_CFC_RTSA = 56, 57, 58
CRAZYCF_RESULT = np.dtype([("total", np.int32), ("printed", np.int32), ("n_prints", np.uint32), ("blocks", np.uint32)])


def graph_to_dict(g):
    """a ctypes graph (oracle's or the product's: same layout) as plain python"""
    n = g.n_nodes
    sb = [g.succ_begin[i] for i in range(n + 1)]
    return {"n_nodes": n, "flags": [g.flags[i] for i in range(n)], "func": [g.func[i] for i in range(n)],
            "succ": [[g.succ[e] for e in range(sb[i], sb[i + 1])] for i in range(n)],
            "calls": [(g.call_node[c], g.call_entry[c]) for c in range(g.n_calls)], "main_func": g.main_func}


def graph_from_dict(d, cls=None):
    """build a ctypes graph from python lists; returns (graph, keepalive)"""
    cls = cls or CfcGraph
    n = d["n_nodes"]
    flags = (C.c_uint8 * n)(*d["flags"])
    func = (C.c_uint16 * n)(*d["func"])
    sb, flat = [0], []
    for sl in d["succ"]:
        flat += list(sl)
        sb.append(len(flat))
    succ_begin = (C.c_uint32 * (n + 1))(*sb)
    succ = (C.c_uint16 * max(1, len(flat)))(*flat)
    nc = len(d["calls"])
    cn = (C.c_uint16 * max(1, nc))(*[c[0] for c in d["calls"]])
    ce = (C.c_uint16 * max(1, nc))(*[c[1] for c in d["calls"]])
    g = cls(n, C.cast(flags, C.POINTER(C.c_uint8)), C.cast(func, C.POINTER(C.c_uint16)),
            C.cast(succ_begin, C.POINTER(C.c_uint32)), C.cast(succ, C.POINTER(C.c_uint16)), nc,
            C.cast(cn, C.POINTER(C.c_uint16)), C.cast(ce, C.POINTER(C.c_uint16)), d["main_func"])
    return g, (flags, func, succ_begin, succ, cn, ce)


def tables_to_dict(t):
    n = t.n_nodes
    sb = [t.succ_begin[i] for i in range(n + 1)]
    return {"n_nodes": n, "n_buffers": t.n_buffers, "sig": list(t.sig[:n]), "sig_diff": list(t.sig_diff[:n]),
            "sig_adj": list(t.sig_adj[:n]), "flags": list(t.flags[:n]),
            "succ": [[t.succ[e] for e in range(sb[i], sb[i + 1])] for i in range(n)],
            "call_pre_adj": list(t.call_pre_adj[:8]), "call_post_adj": list(t.call_post_adj[:8])}


def crazycf_graph():
    g = CfcGraph()
    lib().orc_crazycf_graph(C.byref(g))
    return g


def cfcss_assign(graph):
    """graph: CfcGraph or a dict (graph_from_dict).  Returns CfcTables.  Resets libc's rand() to its unseeded state."""
    keep = None
    if isinstance(graph, dict):
        graph, keep = graph_from_dict(graph)
    t = CfcTables()
    rc = lib().orc_cfcss_assign(C.byref(graph), C.byref(t))
    if rc:
        raise RuntimeError("orc_cfcss_assign: %d" % rc)
    del keep
    return t


def crazycf_plain(seed, size, times):
    res = np.zeros(1, CRAZYCF_RESULT)
    lib().orc_crazycf_plain(C.c_int32(seed), C.c_int32(size), C.c_int32(times), res.ctypes.data_as(C.c_void_p))
    return int(res["total"][0]), int(res["printed"][0]), int(res["n_prints"][0])


def crazycf_batch(params, cfcss=True, faults=None, tables=None):
    """params: (n, 3) int32 rows of (seed, size, timesThroughWhile).  Returns (results structured array, status)."""
    prm = np.ascontiguousarray(params, dtype=np.int32).reshape(-1, 3)
    n = prm.shape[0]
    if tables is None:
        tables = cfcss_assign(crazycf_graph())
    fl = _faults(faults)
    res = np.zeros(n, CRAZYCF_RESULT)
    status = np.zeros(n, np.uint8)
    lib().orc_crazycf_batch(C.byref(tables), C.c_int(1 if cfcss else 0), _p(prm, C.c_int32), C.c_size_t(n),
                            fl.ctypes.data_as(C.c_void_p), C.c_size_t(len(fl)), res.ctypes.data_as(C.c_void_p),
                            _p(status, C.c_uint8))
    return res, status


def crazycf_xmr(params, replicas=3, flags=0, faults=None):
    """crazyCF under -TMR / -DWC (crazycf_xmr.inc): params (n, 3) int32 rows of (seed, size, timesThroughWhile).
    Returns (results structured array, status, stats dict, detected)."""
    prm = np.ascontiguousarray(params, dtype=np.int32).reshape(-1, 3)
    n = prm.shape[0]
    fl = _faults(faults)
    res = np.zeros(n, CRAZYCF_RESULT)
    status = np.zeros(n, np.uint8)
    det = np.zeros(n, np.uint8)
    st = Stats()
    cfg = Cfg(replicas, 0, flags)
    lib().orc_crazycf_xmr(_p(prm, C.c_int32), C.c_size_t(n), C.byref(cfg), fl.ctypes.data_as(C.c_void_p), C.c_size_t(len(fl)),
                          C.byref(st), res.ctypes.data_as(C.c_void_p), _p(status, C.c_uint8), _p(det, C.c_uint8))
    return res, status, st.as_dict(), det


def glibc_rand_seq(seed, k):
    out = np.zeros(k, np.uint32)
    lib().orc_glibc_rand_seq(C.c_uint32(seed & 0xFFFFFFFF), _p(out, C.c_uint32), C.c_size_t(k))
    return out


def ref_crazycf(seed, size):
    """crazyCF.c itself (oracle/_ref), with its srand() argument and its global `size` set; timesThroughWhile stays 10.
    Returns (Total, total-so-far value, number of total-so-far lines)."""
    out = (C.c_int * 3)()
    ref().ref_crazycf(C.c_uint(seed & 0xFFFFFFFF), C.c_int(size), out)
    return int(out[0]), int(out[1]), int(out[2])


def cpu_tmr_mm(f, s, xor_golden):
    """Default-mode CPU-TMR restatement (timing baseline).  Returns (r, error_flag, TMR_ERROR_CNT, syncs)."""
    f = np.ascontiguousarray(f, dtype=np.uint32)
    s = np.ascontiguousarray(s, dtype=np.uint32)
    n = f.shape[-1]
    r = np.empty((n, n), dtype=np.uint32)
    cnt = C.c_uint32(0)
    syncs = C.c_uint64(0)
    err = lib().orc_cpu_tmr_mm(_p(f, C.c_uint32), _p(s, C.c_uint32), _p(r, C.c_uint32), C.c_int(n),
                               C.c_uint32(xor_golden), C.byref(cnt), C.byref(syncs))
    return r, int(err), int(cnt.value), int(syncs.value)


def cpu_tmr_mm_campaign(f, s, xor_golden, faults):
    """Default-mode CPU-TMR restatement under a seeded fault list: run b of the campaign carries the upsets whose item lies in
    matrix b of a virtual batch (one n x n product per run).  Returns the summed TMR_ERROR_CNT and the outcome classes."""
    f = np.ascontiguousarray(f, dtype=np.uint32)
    s = np.ascontiguousarray(s, dtype=np.uint32)
    n = f.shape[-1]
    fl = _faults(faults)
    nruns = int(fl["item"].max() // (n * n)) + 1 if len(fl) else 0
    ne, nc, ns = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    tot = lib().orc_cpu_tmr_mm_campaign(_p(f, C.c_uint32), _p(s, C.c_uint32), C.c_int(n), C.c_uint32(xor_golden),
                                        fl.ctypes.data_as(C.c_void_p), C.c_size_t(len(fl)), C.c_size_t(nruns),
                                        C.byref(ne), C.byref(nc), C.byref(ns))
    return {"runs": nruns, "TMR_ERROR_CNT": int(tot), "error": int(ne.value), "fault_corrected": int(nc.value),
            "success": int(ns.value)}


def cpu_tmr_mm_threads(f, s, xor_golden, nthreads, reps):
    """nthreads x reps default-mode CPU-TMR multiplications in parallel; returns wall seconds."""
    f = np.ascontiguousarray(f, dtype=np.uint32)
    s = np.ascontiguousarray(s, dtype=np.uint32)
    return float(lib().orc_cpu_tmr_mm_threads(_p(f, C.c_uint32), _p(s, C.c_uint32), C.c_int(f.shape[-1]),
                                              C.c_uint32(xor_golden), C.c_int(nthreads), C.c_int(reps)))


# ---------------------------------------------------------------- the reference itself (oracle/_ref)
def ref_quicksort(arr):
    """the reference's own quick_sort (tests/quicksort/quicksort.c:109-129, compiled from where it lies)"""
    a = np.ascontiguousarray(arr, dtype=np.int32).copy()
    ref().ref_quicksort(_p(a, C.c_int32), C.c_int(a.shape[-1]))
    return a


def ref_mm(f, s, golden):
    R = ref()
    n = f.shape[-1]
    fn = getattr(R, "ref_mm%d_run" % n)
    f = np.ascontiguousarray(f, dtype=np.uint32)
    s = np.ascontiguousarray(s, dtype=np.uint32)
    r = np.empty((n, n), dtype=np.uint32)
    err = fn(_p(f, C.c_uint32), _p(s, C.c_uint32), _p(r, C.c_uint32), C.c_uint32(golden))
    return r, int(err)


def ref_sha256(data: bytes) -> bytes:
    R = ref()
    ctx_data = np.zeros(64, np.uint8)
    bitlen = np.zeros(2, np.uint32)
    state = np.zeros(8, np.uint32)
    buf = np.frombuffer(bytes(data) + b"\0", dtype=np.uint8).copy()
    out = np.empty(32, np.uint8)
    R.ref_sha256_hash(_p(ctx_data, C.c_uint8), _p(bitlen, C.c_uint32), _p(state, C.c_uint32), _p(buf, C.c_uint8),
                      C.c_uint32(len(data)), _p(out, C.c_uint8))
    return out.tobytes()


def ref_aes(state: bytes, key: bytes, direction: int):
    R = ref()
    s = np.frombuffer(bytes(state), dtype=np.uint8).copy()
    k = np.frombuffer(bytes(key), dtype=np.uint8).copy()
    R.ref_aes_enc_dec(_p(s, C.c_uint8), _p(k, C.c_uint8), C.c_ubyte(direction))
    return s.tobytes(), k.tobytes()


def ref_crc16(data: bytes) -> int:
    assert len(data) <= 255
    return int(ref().ref_crc16(bytes(data), len(data)))
