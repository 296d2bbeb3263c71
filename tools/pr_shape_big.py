#!/usr/bin/env python3
"""PageRank pull iteration on big graphs of other shapes than Graph500 R-MAT, generated on the device (gdn_rmat_build_ex):
uniform random (a = b = c = 1/4), a milder R-MAT (.45/.22/.22), the LJ-like and Orkut-like recipes scaled up -- where the
small `tools/shapes.py` graphs (67 M edges, 0.17 ms per iteration) mostly measure launch and fill latency.
usage: pr_shape_big.py <scale> <edge factor> <a> <b> <c> [flags=1] ..."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()


def alloc(nbytes):
    p = C.c_void_p()
    _cabi.check(L.gdn_dev_alloc(nbytes, C.byref(p)))
    return p


def run(scale, ef, a, b, c, flags):
    go, gi = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build_ex(scale, ef << scale, a, b, c, graphio.K_RAND_SEED, flags, C.byref(go), C.byref(gi)))
    m, nnz = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(gi, C.byref(m), C.byref(nnz), None, None))
    m, nnz = m.value, nnz.value
    deg, scores0, diff = alloc(4 * m), alloc(4 * m), alloc(8)
    _cabi.check(L.gdn_graph_degrees_dev(go, deg, None))
    L.gdn_graph_free(go)
    init = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
    _cabi.check(L.gdn_dev_upload(scores0, init.ctypes.data_as(C.c_void_p), 4 * m))
    plan = C.c_void_p()
    _cabi.check(L.gdn_pr_plan_create(gi, deg, m, 0, 2, C.byref(plan)))
    ms_ = C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_state_size(plan, C.byref(ms_)))
    ms_ = ms_.value
    state, c0, c1 = alloc(4 * ms_), alloc(4 * ms_ + 16), alloc(4 * ms_ + 16)
    _cabi.check(L.gdn_pr_import_dev(plan, scores0, state, 0.85, None))
    _cabi.check(L.gdn_pr_contrib_dev(plan, state, c0, None))
    bufs, it = [c0, c1], 0
    for _ in range(3):
        _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
        it += 1
    n = 10
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 1, n, None, None))
    for _ in range(n):
        _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
        it += 1
    tot, k = (C.c_double * 2)(0, 0), C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 0, 0, tot, C.byref(k)))
    _cabi.check(L.gdn_pr_plan_check(plan))
    a_ms, b_ms = tot[0] / k.value, tot[1] / k.value
    nb = int(L.gdn_pr_iter_bytes(plan))
    nh, he = C.c_int32(0), C.c_uint64(0)
    _cabi.check(L.gdn_pr_plan_hubs(plan, C.byref(nh), C.byref(he)))
    mt, msrc, me = C.c_int32(0), C.c_int32(0), C.c_uint64(0)
    _cabi.check(L.gdn_pr_plan_mid(plan, C.byref(mt), C.byref(msrc), C.byref(me)))
    L.gdn_pr_plan_free(plan)
    L.gdn_graph_free(gi)
    for p in (deg, scores0, diff, state, c0, c1):
        L.gdn_dev_free(p)
    rec = {"scale": scale, "abc": (a, b, c), "flags": flags, "vertices": m, "state": ms_, "edges": nnz, "A_ms": a_ms, "B_ms": b_ms,
           "ms": a_ms + b_ms, "frac": nb / ((a_ms + b_ms) * 1e-3) / 8e12, "edges_in_record_tiers": (he.value + me.value) / max(nnz, 1),
           "tiers": [nh.value, mt.value, msrc.value]}
    print(json.dumps(rec), flush=True)


args = sys.argv[1:]
while args:
    scale, ef, a, b, c = int(args[0]), int(args[1]), float(args[2]), float(args[3]), float(args[4])
    flags = 1
    args = args[5:]
    if args and args[0].startswith("flags="):
        flags = int(args[0][6:])
        args = args[1:]
    run(scale, ef, a, b, c, flags)
