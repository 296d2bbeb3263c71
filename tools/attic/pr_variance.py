#!/usr/bin/env python3
"""Where the run-to-run spread of the PageRank iteration comes from: in ONE process the plan is built several times (a
dummy allocation of a different size in between, so the arrays land elsewhere) and every plan is timed twice, interleaved
with the others.  A plan whose time stays put while plans differ = placement of its arrays; all plans drifting together =
the device's clocks.  usage: python tools/pr_variance.py [scale] [plans]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
nplans = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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
plans = []
for k in range(nplans):
    dummy = alloc((37 + 101 * k) << 20)  # perturbs where the next plan's arrays go
    scores = alloc(4 * m)
    _cabi.check(L.gdn_dev_upload(scores, init.ctypes.data_as(C.c_void_p), 4 * m))
    plan = C.c_void_p()
    _cabi.check(L.gdn_pr_plan_create(gi, deg, m, 0, 2, C.byref(plan)))
    ms_ = C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_state_size(plan, C.byref(ms_)))
    state, c0, c1, diff = alloc(4 * ms_.value), alloc(4 * ms_.value), alloc(4 * ms_.value), alloc(8)
    _cabi.check(L.gdn_pr_import_dev(plan, scores, state, 0.85, None))
    _cabi.check(L.gdn_pr_contrib_dev(plan, state, c0, None))
    plans.append((plan, state, [c0, c1], diff))


def timed(pl, steps=10):
    plan, state, bufs, diff = pl
    for it in range(2):
        _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 1, steps, None, None))
    for it in range(steps):
        _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
    tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 0, 0, tot, C.byref(n)))
    return tot[0] / n.value, tot[1] / n.value


for rnd in range(3):
    for k, pl in enumerate(plans):
        a, b = timed(pl)
        print("round %d plan %d: A %.3f  B %.3f  sum %.3f ms" % (rnd, k, a, b, a + b), flush=True)
