"""Control-flow signatures: the compile-time half of the reference's CFCSS pass (projects/CFCSS/CFCSS.cpp) behind the C ABI.

Host-only calls (no GPU needed): `assign` hands a control-flow graph to coast_cfcss_assign and returns the signature tables the
kernels check at run time; `crazycf_graph` / `crazycf_tables` are the ones libcoast_hip.so itself uses for tests/crazyCF."""
from __future__ import annotations

import ctypes as C

from . import _lib

FAN_IN, CHECKED, BUFFER, SKIP, RET = 1, 2, 4, 8, 16


def _graph_dict(g) -> dict:
    n = g.n_nodes
    sb = [g.succ_begin[i] for i in range(n + 1)]
    return {"n_nodes": n, "flags": [g.flags[i] for i in range(n)], "func": [g.func[i] for i in range(n)],
            "succ": [[g.succ[e] for e in range(sb[i], sb[i + 1])] for i in range(n)],
            "calls": [(g.call_node[c], g.call_entry[c]) for c in range(g.n_calls)], "main_func": g.main_func}


def _tables_dict(t) -> dict:
    n = t.n_nodes
    sb = [t.succ_begin[i] for i in range(n + 1)]
    return {"n_nodes": n, "n_buffers": t.n_buffers, "sig": list(t.sig[:n]), "sig_diff": list(t.sig_diff[:n]),
            "sig_adj": list(t.sig_adj[:n]), "flags": list(t.flags[:n]),
            "succ": [[t.succ[e] for e in range(sb[i], sb[i + 1])] for i in range(n)],
            "call_pre_adj": list(t.call_pre_adj[:8]), "call_post_adj": list(t.call_post_adj[:8])}


def assign(graph: dict) -> dict:
    """graph: {"n_nodes", "flags": [..], "func": [..], "succ": [[..] per block, terminator operand order], "calls":
    [(calling block, callee entry block)..], "main_func"}.  Returns the tables as a dict (see include/coast_hip.h)."""
    n = graph["n_nodes"]
    flags = (C.c_uint8 * n)(*graph["flags"])
    func = (C.c_uint16 * n)(*graph["func"])
    sb, flat = [0], []
    for sl in graph["succ"]:
        flat += list(sl)
        sb.append(len(flat))
    succ_begin = (C.c_uint32 * (n + 1))(*sb)
    succ = (C.c_uint16 * max(1, len(flat)))(*flat)
    nc = len(graph["calls"])
    cn = (C.c_uint16 * max(1, nc))(*[c[0] for c in graph["calls"]])
    ce = (C.c_uint16 * max(1, nc))(*[c[1] for c in graph["calls"]])
    g = _lib.CoastCfcGraph(n, C.cast(flags, C.POINTER(C.c_uint8)), C.cast(func, C.POINTER(C.c_uint16)),
                           C.cast(succ_begin, C.POINTER(C.c_uint32)), C.cast(succ, C.POINTER(C.c_uint16)), nc,
                           C.cast(cn, C.POINTER(C.c_uint16)), C.cast(ce, C.POINTER(C.c_uint16)), graph["main_func"])
    t = _lib.CoastCfcTables()
    rc = _lib.load().coast_cfcss_assign(C.byref(g), C.byref(t))
    if rc:
        raise _lib.CoastLibraryError("coast_cfcss_assign: error %d" % rc)
    return _tables_dict(t)


def crazycf_graph() -> dict:
    g = _lib.CoastCfcGraph()
    rc = _lib.load().coast_crazycf_graph(C.byref(g))
    if rc:
        raise _lib.CoastLibraryError("coast_crazycf_graph: error %d" % rc)
    return _graph_dict(g)


def crazycf_tables() -> dict:
    t = _lib.CoastCfcTables()
    rc = _lib.load().coast_crazycf_tables(C.byref(t))
    if rc:
        raise _lib.CoastLibraryError("coast_crazycf_tables: error %d" % rc)
    return _tables_dict(t)
