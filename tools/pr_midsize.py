#!/usr/bin/env python3
"""PageRank at the size of BASELINE config 2: the LJ-like stand-in (graphio.LJ_LIKE: 6 M vertices, 70 M edges) --
per knob set the plan's geometry, phase A / phase B per iteration from HIP events, the one-shot gdn_pr solve.
usage: pr_midsize.py "K1=V1,K2=V2" "K3=V3" ...   ("" = defaults; experiment knobs need the var_exp build: GARDENIA_HIP_LIB)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
sets = sys.argv[1:] or [""]
go, gi = C.c_void_p(), C.c_void_p()
if os.environ.get("PR_MIDSIZE_RMAT"):  # an R-MAT graph of that scale instead of the LJ-like stand-in
    _cabi.check(L.gdn_rmat_build(int(os.environ["PR_MIDSIZE_RMAT"]), 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
else:
    r = graphio.LJ_LIKE
    _cabi.check(L.gdn_rmat_build_ex(r["scale"], r["n_edges"], *r["abc"], graphio.K_RAND_SEED, r["flags"], C.byref(go), C.byref(gi)))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(gi, C.byref(m), C.byref(nnz), None, None))
m, nnz = m.value, nnz.value


def alloc(nbytes):
    p = C.c_void_p()
    _cabi.check(L.gdn_dev_alloc(nbytes, C.byref(p)))
    return p


deg = alloc(4 * m)
_cabi.check(L.gdn_graph_degrees_dev(go, deg, None))
h_rp, h_ci = np.empty(m + 1, np.uint64), np.empty(nnz, np.int32)
_cabi.check(L.gdn_graph_download(gi, h_rp.ctypes.data_as(C.c_void_p), h_ci.ctypes.data_as(C.c_void_p)))
h_deg = np.empty(m, np.int32)
_cabi.check(L.gdn_dev_download(h_deg.ctypes.data_as(C.c_void_p), deg, 4 * m))
L.gdn_graph_free(go)
init = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
print("%s: %d vertices, %d edges" % ("R-MAT-" + os.environ["PR_MIDSIZE_RMAT"] if os.environ.get("PR_MIDSIZE_RMAT") else "LJ-like stand-in", m, nnz))
for spec in sets:
    env = dict(kv.split("=") for kv in spec.split(",") if kv)
    for k, v in env.items():
        _cabi.check(L.gdn_option_set(k.encode(), v.encode()))
    plan = C.c_void_p()
    _cabi.check(L.gdn_pr_plan_create(gi, deg, m, 0, 2, C.byref(plan)))
    lay, lg, nb, ms_ = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_layout(plan, C.byref(lay), C.byref(lg)))
    _cabi.check(L.gdn_pr_plan_bins(plan, C.byref(nb)))
    _cabi.check(L.gdn_pr_plan_state_size(plan, C.byref(ms_)))
    scores, state, c0, c1, diff = alloc(4 * m), alloc(4 * ms_.value), alloc(4 * ms_.value + 16), alloc(4 * ms_.value + 16), alloc(8)
    _cabi.check(L.gdn_dev_upload(scores, init.ctypes.data_as(C.c_void_p), 4 * m))
    _cabi.check(L.gdn_pr_import_dev(plan, scores, state, 0.85, None))
    _cabi.check(L.gdn_pr_contrib_dev(plan, state, c0, None))
    bufs = [c0, c1]
    best = None
    it = 0
    for batch in range(4):
        _cabi.check(L.gdn_pr_plan_kernel_time(plan, 1, 20, None, None))
        for _ in range(20):
            _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
            it += 1
        tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
        _cabi.check(L.gdn_pr_plan_kernel_time(plan, 0, 0, tot, C.byref(n)))
        cur = (tot[0] / n.value, tot[1] / n.value)
        if batch and (best is None or sum(cur) < sum(best)):
            best = cur
    nbytes = int(L.gdn_pr_iter_bytes(plan))
    L.gdn_pr_plan_free(plan)
    for p in (scores, state, c0, c1, diff):
        L.gdn_dev_free(p)
    solves = []
    for _ in range(4):
        sc = init.copy()
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_pr(m, nnz, h_rp.ctypes.data_as(C.c_void_p), h_ci.ctypes.data_as(C.c_void_p), h_deg.ctypes.data_as(C.c_void_p),
                             sc.ctypes.data_as(C.c_void_p), C.c_float(0.85), C.c_double(1e-4), 100, C.byref(st)))
        solves.append((st.solve_ms, st.prep_ms, st.iterations))
    b = min(solves)
    print("%-48s log_blk %d bins %d | A %.4f B %.4f sum %.4f ms = %.3f of the roofline | gdn_pr: solve %.2f ms + prep %.2f, %d iterations" % (
        spec or "(defaults)", lg.value, nb.value, best[0], best[1], sum(best), nbytes / (sum(best) * 1e-3) / 8e12, b[0], b[1], b[2]), flush=True)
    for k in env:
        _cabi.check(L.gdn_option_set(k.encode(), None))
