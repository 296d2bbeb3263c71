#!/usr/bin/env python3
"""Follow-up to pr_variance.py: is it the plan's arrays or the caller's state vectors whose placement moves the iteration
time?  Three plans x three sets of state vectors, every pairing timed.  usage: python tools/pr_variance2.py [scale]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
go, gi = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
m = C.c_int32()
_cabi.check(L.gdn_graph_info(gi, C.byref(m), None, None, None))
m = m.value


def alloc(nbytes):
    p = C.c_void_p()
    _cabi.check(L.gdn_dev_alloc(nbytes, C.byref(p)))
    return p


deg = alloc(4 * m)
_cabi.check(L.gdn_graph_degrees_dev(go, deg, None))
L.gdn_graph_free(go)
import numpy as np
init = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
scores = alloc(4 * m)
_cabi.check(L.gdn_dev_upload(scores, init.ctypes.data_as(C.c_void_p), 4 * m))
plans, states = [], []
ms = None
for k in range(3):
    alloc((53 + 67 * k) << 20)
    plan = C.c_void_p()
    _cabi.check(L.gdn_pr_plan_create(gi, deg, m, 0, 2, C.byref(plan)))
    ms_ = C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_state_size(plan, C.byref(ms_)))
    ms = ms_.value
    plans.append(plan)
for k in range(3):
    alloc((11 + 29 * k) << 20)
    states.append((alloc(4 * ms), alloc(4 * ms), alloc(4 * ms), alloc(8)))


def timed(plan, st, steps=10):
    state, c0, c1, diff = st
    _cabi.check(L.gdn_pr_import_dev(plan, scores, state, 0.85, None))
    _cabi.check(L.gdn_pr_contrib_dev(plan, state, c0, None))
    bufs = [c0, c1]
    for it in range(2):
        _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 1, steps, None, None))
    for it in range(steps):
        _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
    tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 0, 0, tot, C.byref(n)))
    return tot[0] / n.value, tot[1] / n.value


for pi, plan in enumerate(plans):
    for si, st in enumerate(states):
        a, b = timed(plan, st)
        print("plan %d state %d: A %.3f  B %.3f  sum %.3f ms" % (pi, si, a, b, a + b), flush=True)
