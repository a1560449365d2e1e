"""The protected kernels under their reference names, with the reference's data contract (host buffers in/out).

    matrix_multiply   tests/mm_common/mm_common_tmr.c:3      void matrix_multiply(f[][side], s[][side], r[][side])
    sha256_hash       tests/sha256_common/sha256_common_tmr.c:101
    aes_enc_dec       tests/aes/TI_aes_128.h:42              state and key updated in place
    crc16             tests/crc16/crc16.c:21

Each call is one H2D copy, one protected launch on the GPU and one D2H copy through libcoast_hip.so's host shims.
TMR never reports (it corrects and counts); DWC raises FaultDetectedDWC where the reference would call
FAULT_DETECTED_DWC() and abort (synchronization.cpp:1299-1302).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .engine import DWC, TMR, XmrConfig


class FaultDetectedDWC(RuntimeError):
    """FAULT_DETECTED_DWC analogue."""


def _cfg(cfg):
    return (cfg or XmrConfig()).c()


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))


def host_stats(reset: bool = False) -> dict:
    """TMR_ERROR_CNT / __SYNC_COUNT as accumulated by the single-call shims."""
    st = _lib.CoastStats()
    _check(_lib.load().coast_host_stats(C.byref(st), int(reset)), "coast_host_stats")
    return {"errors_corrected": int(st.errors_corrected), "sync_count": int(st.sync_count),
            "dwc_detected": int(st.dwc_detected), "launches": int(st.launches)}


def _dwc_check(before):
    if host_stats()["dwc_detected"] > before:
        raise FaultDetectedDWC("DWC compare failed")


def matrix_multiply(f_matrix, s_matrix, cfg: XmrConfig | None = None):
    f = np.ascontiguousarray(f_matrix, dtype=np.uint32)
    s = np.ascontiguousarray(s_matrix, dtype=np.uint32)
    side = f.shape[0]
    assert f.shape == (side, side) == s.shape
    r = np.empty_like(f)
    cc = _cfg(cfg)
    before = host_stats()["dwc_detected"]
    _check(_lib.load().coast_matrix_multiply_host(f.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p),
                                                  r.ctypes.data_as(C.c_void_p), side, C.byref(cc)),
           "coast_matrix_multiply_host")
    _dwc_check(before)
    return r


def sha256_hash(data: bytes, cfg: XmrConfig | None = None) -> bytes:
    buf = np.frombuffer(bytes(data) + b"\0", dtype=np.uint8).copy()
    out = np.empty(32, dtype=np.uint8)
    cc = _cfg(cfg)
    before = host_stats()["dwc_detected"]
    _check(_lib.load().coast_sha256_host(buf.ctypes.data_as(C.c_void_p), len(data), out.ctypes.data_as(C.c_void_p),
                                         None, C.byref(cc)), "coast_sha256_host")
    _dwc_check(before)
    return out.tobytes()


def aes_enc_dec(state: bytes, key: bytes, direction: int, cfg: XmrConfig | None = None):
    """Returns (state, key) after the call -- the reference mutates both buffers."""
    s = np.frombuffer(bytes(state), dtype=np.uint8).copy()
    k = np.frombuffer(bytes(key), dtype=np.uint8).copy()
    cc = _cfg(cfg or XmrConfig(DWC))
    before = host_stats()["dwc_detected"]
    _check(_lib.load().coast_aes_enc_dec_host(s.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p),
                                              int(direction) & 0xFF, C.byref(cc)), "coast_aes_enc_dec_host")
    _dwc_check(before)
    return s.tobytes(), k.tobytes()


def crc16(data: bytes, cfg: XmrConfig | None = None) -> int:
    buf = np.frombuffer(bytes(data) + b"\0", dtype=np.uint8).copy()
    out = np.zeros(1, dtype=np.uint16)
    cc = _cfg(cfg)
    before = host_stats()["dwc_detected"]
    _check(_lib.load().coast_crc16_host(buf.ctypes.data_as(C.c_void_p), len(data), out.ctypes.data_as(C.c_void_p),
                                        C.byref(cc)), "coast_crc16_host")
    _dwc_check(before)
    return int(out[0])


__all__ = ["matrix_multiply", "sha256_hash", "aes_enc_dec", "crc16", "host_stats", "FaultDetectedDWC", "TMR", "DWC"]
