#!/usr/bin/env python3
"""Does phase A's placement spread follow the ALLOCATION or the process?  One process, RMAT-27: K plans created one after the
other and all kept alive (so every one sits in other memory), each timed over three batches of 10 iterations, round-robin twice.
usage: pr_replan.py [scale] [K]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
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
plans = []
for k in range(K):
    plan = C.c_void_p()
    _cabi.check(L.gdn_pr_plan_create(gi, deg, m, 0, 2, C.byref(plan)))
    ms_ = C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_state_size(plan, C.byref(ms_)))
    ms_ = ms_.value
    state, c0, c1 = alloc(4 * ms_), alloc(4 * ms_), alloc(4 * ms_)
    _cabi.check(L.gdn_pr_import_dev(plan, scores0, state, 0.85, None))
    _cabi.check(L.gdn_pr_contrib_dev(plan, state, c0, None))
    plans.append((plan, state, [c0, c1]))
for rnd in range(2):
    for k, (plan, state, bufs) in enumerate(plans):
        it = 0
        for _ in range(3):
            _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
            it += 1
        res = []
        for batch in range(3):
            _cabi.check(L.gdn_pr_plan_kernel_time(plan, 1, 10, None, None))
            for _ in range(10):
                _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
                it += 1
            tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
            _cabi.check(L.gdn_pr_plan_kernel_time(plan, 0, 0, tot, C.byref(n)))
            res.append((tot[0] / n.value, tot[1] / n.value))
        print("round %d plan %d: " % (rnd, k) + "  ".join("A %.3f B %.3f = %.3f" % (a, b, a + b) for a, b in res), flush=True)
