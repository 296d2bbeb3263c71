#!/usr/bin/env python3
"""Slow and fast placements of ONE PageRank plan in ONE process, for the per-channel counter passes of tools/pr_channels.sh
(VERDICT r4 item 2).  RMAT-<scale> squished blocked plan without the placement search; per configuration 1 untimed + 3 timed
iterations, then `vals` (what phase A's time follows) and the record streams (phase B) move into fresh allocations behind a
spacer that stays allocated, so every configuration lives in other memory.  Prints one line per configuration; the profiler's
dispatch order is the configuration order (4 pb_expand_kernel<0> / pb_accumulate_kernel<PrOp, 0> dispatches each).
usage: pr_channels.py [scale] [configs]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
configs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
os.environ["GDN_PR_PLACE"] = "0"
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
state, c0, c1 = alloc(4 * ms_), alloc(4 * ms_ + 16), alloc(4 * ms_ + 16)
_cabi.check(L.gdn_pr_import_dev(plan, scores0, state, 0.85, None))
_cabi.check(L.gdn_pr_contrib_dev(plan, state, c0, None))
bufs = [c0, c1]
it = 0


def pulls(n):
    global it
    for _ in range(n):
        _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
        it += 1


def timed(n=3):
    pulls(1)
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 1, n, None, None))
    pulls(n)
    tot, k = (C.c_double * 2)(0, 0), C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 0, 0, tot, C.byref(k)))
    return tot[0] / k.value, tot[1] / k.value


spacers = []
for c in range(configs):
    a, b = timed()
    print("config %d: A %.3f B %.3f = %.3f" % (c, a, b, a + b), flush=True)
    if c + 1 < configs:
        spacers.append(alloc((3 << 30) + (c << 21)))  # stays allocated: the next copies land elsewhere
        _cabi.check(L.gdn_pr_plan_move(plan, 1 | 16 | 32))
