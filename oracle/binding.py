"""ctypes binding of oracle/_build/libgardenia_oracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only
(never by gardenia_amd/).  See oracle/gardenia_oracle.cc for the reference citations.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgardenia_oracle.so")
REF_DIR = os.path.join(_HERE, "_ref")

_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile the restatement (and, when /root/reference exists, oracle/_ref)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "gardenia_oracle.cc"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if os.path.isdir("/root/reference/src") and not os.path.exists(os.path.join(REF_DIR, "ref_bc")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_num_threads.restype = C.c_int
        L.orc_bfs_serial.argtypes = [C.c_int32, _u64p, _i32p, C.c_int32, _i32p]
        L.orc_bfs_serial.restype = None
        L.orc_bfs_topdown.argtypes = [C.c_int32, _u64p, _i32p, C.c_int32, _i32p]
        L.orc_bfs_topdown.restype = C.c_int
        L.orc_bfs_beamer.argtypes = [C.c_int32, _u64p, _i32p, _u64p, _i32p, C.c_int32, _i32p]
        L.orc_bfs_beamer.restype = C.c_int
        L.orc_bfs_verify.argtypes = [C.c_int32, _u64p, _i32p, C.c_int32, _i32p]
        L.orc_bfs_verify.restype = C.c_int
        L.orc_pr.argtypes = [C.c_int32, _u64p, _i32p, _i32p, _f32p, C.c_float, C.c_double,
                             C.c_int, C.c_void_p]
        L.orc_pr.restype = C.c_int
        L.orc_pr_iterate.argtypes = [C.c_int32, _u64p, _i32p, _i32p, _f32p, C.c_float, C.c_int,
                                     C.c_int32, C.c_int32]
        L.orc_pr_iterate.restype = C.c_double
        L.orc_pr_verify_error.argtypes = [C.c_int32, _u64p, _i32p, _f32p, C.c_float]
        L.orc_pr_verify_error.restype = C.c_double
        L.orc_spmv.argtypes = [C.c_int32, _u64p, _i32p, _f32p, _f32p, _f32p]
        L.orc_spmv.restype = None
        L.orc_spmv_max_rel_error.argtypes = [_f32p, _f32p, C.c_int64]
        L.orc_spmv_max_rel_error.restype = C.c_float
        L.orc_bytes_per_spmv.argtypes = [C.c_int64, C.c_int64]
        L.orc_bytes_per_spmv.restype = C.c_uint64
        L.orc_sssp_dijkstra.argtypes = [C.c_int32, _u64p, _i32p, _i32p, C.c_int32, _i32p]
        L.orc_sssp_dijkstra.restype = None
        L.orc_sssp_delta.argtypes = [C.c_int32, _u64p, _i32p, _i32p, C.c_int32, C.c_int32, _i32p]
        L.orc_sssp_delta.restype = None
        L.orc_sssp_verify.argtypes = [C.c_int32, _u64p, _i32p, _i32p, C.c_int32, _i32p]
        L.orc_sssp_verify.restype = C.c_int
        L.orc_cc_sv.argtypes = [C.c_int32, _u64p, _i32p, _i32p]
        L.orc_cc_sv.restype = C.c_int
        L.orc_cc_afforest.argtypes = [C.c_int32, _u64p, _i32p, C.c_void_p, C.c_void_p, _i32p]
        L.orc_cc_afforest.restype = None
        L.orc_cc_verify.argtypes = [C.c_int32, _u64p, _i32p, _i32p]
        L.orc_cc_verify.restype = C.c_int
        L.orc_sample_frequent_element.argtypes = [C.c_int32, _i32p, C.c_int64]
        L.orc_sample_frequent_element.restype = C.c_int32
        L.orc_pr_delta.argtypes = [C.c_int32, _u64p, _i32p, _u64p, _i32p, _i32p, _f32p, C.c_float, C.c_double, C.c_float,
                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_pr_delta.restype = C.c_int
        L.orc_bc.argtypes = [C.c_int32, _u64p, _i32p, C.c_int32, _f32p, C.c_void_p, C.c_void_p]
        L.orc_bc.restype = C.c_int
        L.orc_bc_verify.argtypes = [C.c_int32, _u64p, _i32p, C.c_int32, _f32p]
        L.orc_bc_verify.restype = C.c_int
        L.orc_tc_orient.argtypes = [C.c_int32, _u64p, _i32p, _u64p, C.c_void_p]
        L.orc_tc_orient.restype = C.c_uint64
        L.orc_tc.argtypes = [C.c_int32, _u64p, _i32p]
        L.orc_tc.restype = C.c_uint64
        _lib = L
    return _lib


def _g(g):
    return (np.ascontiguousarray(g.rowptr, dtype=np.uint64),
            np.ascontiguousarray(g.colidx, dtype=np.int32))


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    L = lib()
    L.orc_set_num_threads.argtypes = [C.c_int]
    L.orc_set_num_threads.restype = None
    L.orc_set_num_threads(int(n))


def bfs_serial(g, source):
    rp, ci = _g(g)
    dist = np.empty(g.m, dtype=np.int32)
    lib().orc_bfs_serial(g.m, rp, ci, source, dist)
    return dist


def bfs_topdown(g, source):
    rp, ci = _g(g)
    dist = np.empty(g.m, dtype=np.int32)
    it = lib().orc_bfs_topdown(g.m, rp, ci, source, dist)
    return dist, it


def bfs_beamer(g_out, g_in, source):
    rp, ci = _g(g_out)
    irp, ici = _g(g_in)
    dist = np.empty(g_out.m, dtype=np.int32)
    it = lib().orc_bfs_beamer(g_out.m, rp, ci, irp, ici, source, dist)
    return dist, it


def bfs_verify(g, source, dist) -> bool:
    rp, ci = _g(g)
    return bool(lib().orc_bfs_verify(g.m, rp, ci, source, np.ascontiguousarray(dist, np.int32)))


def pr(g_in, out_degree, damping=0.85, epsilon=1e-4, max_iter=100, scores=None):
    rp, ci = _g(g_in)
    m = g_in.m
    if scores is None:
        scores = np.full(m, np.float32(1.0) / np.float32(m), dtype=np.float32)
    trace = np.zeros(max_iter, dtype=np.float64)
    it = lib().orc_pr(m, rp, ci, np.ascontiguousarray(out_degree, np.int32), scores,
                      damping, epsilon, max_iter, trace.ctypes.data_as(C.c_void_p))
    return scores, it, trace[:min(it, max_iter)]


def pr_iterate(g_in, out_degree, scores, iters, damping=0.85, row_lo=0, row_hi=None):
    rp, ci = _g(g_in)
    row_hi = g_in.m if row_hi is None else row_hi
    err = lib().orc_pr_iterate(g_in.m, rp, ci, np.ascontiguousarray(out_degree, np.int32), scores,
                               damping, iters, row_lo, row_hi)
    return scores, float(err)


def pr_delta(g_in, g_out, damping=0.85, epsilon=1e-4, epsilon2=1e-3, max_iter=100, push_div=10, scores=None):
    """src/pr/omp_delta.cc:52 (push_div 10) / src/pr/delta.cu:140 (push_div 8).
    Returns (scores, iterations, trace) with trace = dict(diff, items, mode) per iteration."""
    m = g_in.m
    if scores is None:
        scores = np.full(m, np.float32(1.0) / np.float32(m), dtype=np.float32)
    irp, ici = _g(g_in)
    orp, oci = _g(g_out)
    deg = np.ascontiguousarray(np.diff(g_out.rowptr.astype(np.int64)), np.int32)
    td, ti, tm = np.zeros(max_iter), np.zeros(max_iter, np.int32), np.zeros(max_iter, np.int32)
    it = lib().orc_pr_delta(m, irp, ici, orp, oci, deg, scores, damping, epsilon, epsilon2, max_iter, push_div,
                            td.ctypes.data_as(C.c_void_p), ti.ctypes.data_as(C.c_void_p), tm.ctypes.data_as(C.c_void_p))
    return scores, it, dict(diff=td[:it], items=ti[:it], mode=tm[:it])


def pr_verify_error(g_out, scores, damping=0.85) -> float:
    rp, ci = _g(g_out)
    return float(lib().orc_pr_verify_error(g_out.m, rp, ci,
                                           np.ascontiguousarray(scores, np.float32), damping))


def spmv(g_in, Ax, x, y0):
    rp, ci = _g(g_in)
    y = np.array(y0, dtype=np.float32, copy=True)
    lib().orc_spmv(g_in.m, rp, ci, np.ascontiguousarray(Ax, np.float32),
                   np.ascontiguousarray(x, np.float32), y)
    return y


def spmv_max_rel_error(a, b) -> float:
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return float(lib().orc_spmv_max_rel_error(a, b, a.size))


def sssp_dijkstra(g, weight, source):
    rp, ci = _g(g)
    dist = np.empty(g.m, dtype=np.int32)
    lib().orc_sssp_dijkstra(g.m, rp, ci, np.ascontiguousarray(weight, np.int32), source, dist)
    return dist


def sssp_delta(g, weight, source, delta=1):
    rp, ci = _g(g)
    dist = np.empty(g.m, dtype=np.int32)
    lib().orc_sssp_delta(g.m, rp, ci, np.ascontiguousarray(weight, np.int32), source, delta, dist)
    return dist


def cc_sv(g):
    rp, ci = _g(g)
    comp = np.empty(g.m, dtype=np.int32)
    it = lib().orc_cc_sv(g.m, rp, ci, comp)
    return comp, it


def cc_afforest(g, g_in=None):
    rp, ci = _g(g)
    comp = np.empty(g.m, dtype=np.int32)
    if g_in is None:
        lib().orc_cc_afforest(g.m, rp, ci, None, None, comp)
    else:
        irp, ici = _g(g_in)
        lib().orc_cc_afforest(g.m, rp, ci, irp.ctypes.data_as(C.c_void_p),
                              ici.ctypes.data_as(C.c_void_p), comp)
    return comp


def cc_verify(g, comp) -> bool:
    rp, ci = _g(g)
    return bool(lib().orc_cc_verify(g.m, rp, ci, np.ascontiguousarray(comp, np.int32)))


def tc_orient(g):
    from gardenia_amd.graphio import CSR
    rp, ci = _g(g)
    nrp = np.zeros(g.m + 1, dtype=np.uint64)
    n = lib().orc_tc_orient(g.m, rp, ci, nrp, None)
    nci = np.empty(int(n), dtype=np.int32)
    lib().orc_tc_orient(g.m, rp, ci, nrp, nci.ctypes.data_as(C.c_void_p))
    return CSR(g.m, nrp, nci)


def tc(g_dag) -> int:
    rp, ci = _g(g_dag)
    return int(lib().orc_tc(g_dag.m, rp, ci))


def bc(g, source, scores=None):
    """src/bc/omp_base.cc BCSolver from one source: returns (scores, levels, depths, path_counts)."""
    sc = np.zeros(g.m, np.float32) if scores is None else np.array(scores, dtype=np.float32)
    depths = np.empty(g.m, np.int32)
    pcs = np.empty(g.m, np.int32)
    lv = lib().orc_bc(g.m, np.ascontiguousarray(g.rowptr, np.uint64), np.ascontiguousarray(g.colidx, np.int32), int(source), sc,
                      depths.ctypes.data_as(C.c_void_p), pcs.ctypes.data_as(C.c_void_p))
    return sc, lv, depths, pcs


def bc_verify(g, source, scores) -> bool:
    """src/bc/verifier.cc criterion (abs 1e-4 + rel 1e-4 against a serial Brandes)."""
    return bool(lib().orc_bc_verify(g.m, np.ascontiguousarray(g.rowptr, np.uint64), np.ascontiguousarray(g.colidx, np.int32),
                                    int(source), np.ascontiguousarray(scores, np.float32)))
