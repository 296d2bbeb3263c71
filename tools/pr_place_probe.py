#!/usr/bin/env python3
"""Which array's placement moves a PageRank iteration?  One plan on RMAT-<scale>; one array group at a time is copied into a
fresh allocation (gdn_pr_plan_move) and phases A / B are timed before and after.  usage: pr_place_probe.py [scale] [rounds]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
go, gi = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(gi, C.byref(m), C.byref(nnz), None, None))
m, nnz = m.value, nnz.value


def alloc(nbytes):
    p = C.c_void_p()
    _cabi.check(L.gdn_dev_alloc(nbytes, C.byref(p)))
    return p


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
state, c0, c1 = alloc(4 * ms_), alloc(4 * ms_), alloc(4 * ms_)
_cabi.check(L.gdn_pr_import_dev(plan, scores0, state, 0.85, None))
_cabi.check(L.gdn_pr_contrib_dev(plan, state, c0, None))
bufs = [c0, c1]
it = 0


def timed(n=8):
    global it
    for _ in range(2):
        _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
        it += 1
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 1, n, None, None))
    for _ in range(n):
        _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
        it += 1
    tot, k = (C.c_double * 2)(0, 0), C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 0, 0, tot, C.byref(k)))
    return tot[0] / k.value, tot[1] / k.value


a, b = timed()
print("start: A %.3f B %.3f = %.3f" % (a, b, a + b), flush=True)
names = {1: "vals", 2: "U", 4: "G", 8: "V", 16: "hub records", 32: "mid records", 64: "tables"}
for r in range(rounds):
    for bit in (1, 2, 4, 8, 16, 32, 64):
        _cabi.check(L.gdn_pr_plan_move(plan, bit))
        a2, b2 = timed()
        print("round %d moved %-12s: A %.3f (%+.3f) B %.3f (%+.3f) = %.3f" % (r, names[bit], a2, a2 - a, b2, b2 - b, a2 + b2), flush=True)
        a, b = a2, b2
