#!/usr/bin/env python3
"""A/B of the allocation stagger (option GDN_ALLOC_STAGGER, gdn_common.hpp) on the PageRank plan: ONE process builds
`reps` plans per granule, interleaved (g0, g1, ..., g0, g1, ...), and times every plan in turn, three rounds.  A granule
whose plans are consistently faster than granule 0 (= plain hipMalloc bases) is a placement remedy; plans of one granule
that differ as much as plans of different granules = no remedy.
usage: python tools/pr_stagger.py [scale] [reps] [granule ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
granules = [int(a) for a in sys.argv[3:]] or [0, 256, 4352, 69888, 2101504]
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
plans = []
for r in range(reps):
    for g in granules:
        _cabi.check(L.gdn_option_set(b"GDN_ALLOC_STAGGER", str(g).encode()))
        plan = C.c_void_p()
        _cabi.check(L.gdn_pr_plan_create(gi, deg, m, 0, 2, C.byref(plan)))
        ms_ = C.c_int32(0)
        _cabi.check(L.gdn_pr_plan_state_size(plan, C.byref(ms_)))
        state, c0, c1, diff = alloc(4 * ms_.value), alloc(4 * ms_.value), alloc(4 * ms_.value), alloc(8)
        _cabi.check(L.gdn_pr_import_dev(plan, scores, state, 0.85, None))
        _cabi.check(L.gdn_pr_contrib_dev(plan, state, c0, None))
        plans.append((g, r, plan, state, [c0, c1], diff))
_cabi.check(L.gdn_option_set(b"GDN_ALLOC_STAGGER", b"0"))


def timed(pl, steps=10):
    _, _, plan, state, bufs, diff = pl
    for it in range(2):
        _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 1, steps, None, None))
    for it in range(steps):
        _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
    tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 0, 0, tot, C.byref(n)))
    return tot[0] / n.value, tot[1] / n.value


best = {}
for rnd in range(3):
    for pl in plans:
        a, b = timed(pl)
        key = (pl[0], pl[1])
        best[key] = min(best.get(key, 1e9), a + b)
        print("round %d granule %8d plan %d: A %.3f  B %.3f  sum %.3f ms" % (rnd, pl[0], pl[1], a, b, a + b), flush=True)
print("--- best of 3 rounds per plan")
for g in granules:
    print("granule %8d: %s" % (g, "  ".join("%.3f" % best[(g, r)] for r in range(reps))))
