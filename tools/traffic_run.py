#!/usr/bin/env python3
"""One workload of the bench line, built once and solved <reps> times, for the HBM counter passes of tools/traffic.sh:
   python3 tools/traffic_run.py <bfs|sssp_unit|sssp_u255|cc|cc_out|tc|spmv> <scale> <reps>
The counter totals of two runs with different <reps> differ by the traffic of the extra solves -- graph and plan builds
cancel -- which is how tools/traffic_summary.py gets bytes per solve without marker traces (gpurun refuses --pmc with them)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
what, scale, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m, nnz = C.c_int32(), C.c_uint64()


def dev_alloc(nbytes):
    p = C.c_void_p()
    _cabi.check(L.gdn_dev_alloc(nbytes, C.byref(p)))
    return p


def dev(a):
    p = dev_alloc(a.nbytes)
    _cabi.check(L.gdn_dev_upload(p, a.ctypes.data_as(C.c_void_p), a.nbytes))
    return p


def first_sources(go, n):
    deg = dev_alloc(4 * m.value)
    _cabi.check(L.gdn_graph_degrees_dev(go, deg, None))
    h = np.empty(1 << 16, np.int32)
    _cabi.check(L.gdn_dev_download(h.ctypes.data_as(C.c_void_p), deg, 4 * (1 << 16)))
    return np.nonzero(h > 0)[0][:n].tolist()


ms = []
if what.startswith("bfs"):  # "bfs" = the first non-isolated source, "bfs:1" the second (RMAT-27: source 5, the slow one), ...
    go, gi = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
    _cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None))
    k = int(what.split(":")[1]) if ":" in what else 0
    src = first_sources(go, k + 1)[k]
    dist = dev_alloc(4 * m.value)
    plan = C.c_void_p()
    _cabi.check(L.gdn_bfs_plan_create(go, gi, 1, C.byref(plan)))
    for _ in range(reps):
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_bfs_run(plan, int(src), dist, C.byref(st)))
        ms.append(st.solve_ms)
elif what in ("sssp_unit", "sssp_u255"):
    go = C.c_void_p()
    _cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
    _cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None))
    src = first_sources(go, 1)[0]
    rng = np.random.default_rng(5)
    w = dev(np.ones(nnz.value, np.int32) if what == "sssp_unit" else rng.integers(1, 256, nnz.value, dtype=np.int32))
    delta = 1 if what == "sssp_unit" else 16
    dist = dev_alloc(4 * m.value)
    plan = C.c_void_p()
    _cabi.check(L.gdn_sssp_plan_create(go, w, 1, C.byref(plan)))
    for _ in range(reps):
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_sssp_run(plan, int(src), delta, dist, C.byref(st)))
        ms.append(st.solve_ms)
elif what in ("cc", "cc_out"):
    go, gi = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
    _cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None))
    comp = dev_alloc(4 * m.value)
    for _ in range(reps):
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_cc_dev(go, gi if what == "cc" else None, comp, C.byref(st)))
        ms.append(st.solve_ms)
elif what == "tc":
    go, gs = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
    _cabi.check(L.gdn_graph_symmetrize(go, C.byref(gs)))
    L.gdn_graph_free(go)
    _cabi.check(L.gdn_graph_info(gs, C.byref(m), C.byref(nnz), None, None))
    plan = C.c_void_p()
    _cabi.check(L.gdn_tc_plan_create(gs, 0, C.byref(plan)))
    for _ in range(reps):
        total, st = C.c_uint64(0), _cabi.GdnStats()
        _cabi.check(L.gdn_tc_plan_count(plan, C.byref(total), C.byref(st)))
        ms.append(st.solve_ms)
elif what == "spmv":
    gi = C.c_void_p()
    _cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, None, C.byref(gi)))
    _cabi.check(L.gdn_graph_info(gi, C.byref(m), C.byref(nnz), None, None))
    rng = np.random.default_rng(7)
    Ax, x, y = dev(rng.random(nnz.value, dtype=np.float32)), dev(rng.random(m.value, dtype=np.float32)), dev(np.zeros(m.value, np.float32))
    plan = C.c_void_p()
    _cabi.check(L.gdn_spmv_plan_create(gi, Ax, 1, C.byref(plan)))
    import time
    one = np.empty(1, np.float32)
    for _ in range(reps):
        t0 = time.perf_counter()
        _cabi.check(L.gdn_spmv_dev(plan, Ax, x, y, None))
        _cabi.check(L.gdn_dev_download(one.ctypes.data_as(C.c_void_p), y, 4))  # (synchronises)
        ms.append((time.perf_counter() - t0) * 1e3)
else:
    raise SystemExit("unknown workload " + what)
print("traffic_run %s scale %d: %d solves, vertices %d edges %d, solve ms %s" % (what, scale, reps, m.value, nnz.value, ["%.3f" % x for x in ms[-3:]]))
