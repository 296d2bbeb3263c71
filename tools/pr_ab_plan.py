#!/usr/bin/env python3
"""A/B of a PLAN-BUILD knob: python tools/pr_ab_plan.py NAME VAL_A VAL_B [scale] [rounds]
One process, one graph; every round builds a plan under each value (placement search off, so a round costs ~0.2 s of
build), times 12 iterations with the plan's own per-kernel events and checks that both variants return the same bits."""
import ctypes as C
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gardenia_amd import _cabi, graphio
import numpy as np

L = _cabi.lib()
name, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
scale = int(sys.argv[4]) if len(sys.argv) > 4 else 27
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 4
_cabi.check(L.gdn_option_set(b"GDN_PR_PLACE", b"0"))
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
init = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
scores = alloc(4 * m)
out = np.empty(m, np.float32)
res = {va: [], vb: []}
crc = {}
for rnd in range(rounds):
    for v in (va, vb) if rnd % 2 == 0 else (vb, va):
        _cabi.check(L.gdn_option_set(name.encode(), v.encode()))
        _cabi.check(L.gdn_dev_upload(scores, init.ctypes.data_as(C.c_void_p), 4 * m))
        plan = C.c_void_p()
        _cabi.check(L.gdn_pr_plan_create(gi, deg, m, 0, 2, C.byref(plan)))
        ms_ = C.c_int32(0)
        _cabi.check(L.gdn_pr_plan_state_size(plan, C.byref(ms_)))
        state, c0, c1, diff = alloc(4 * ms_.value), alloc(4 * ms_.value), alloc(4 * ms_.value), alloc(8)
        _cabi.check(L.gdn_pr_import_dev(plan, scores, state, 0.85, None))
        _cabi.check(L.gdn_pr_contrib_dev(plan, state, c0, None))
        bufs = [c0, c1]
        steps = 12
        for it in range(2):
            _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
        _cabi.check(L.gdn_pr_plan_kernel_time(plan, 1, steps, None, None))
        for it in range(steps):
            _cabi.check(L.gdn_pr_pull_dev(plan, bufs[it & 1], state, bufs[(it + 1) & 1], diff, 0.85, None))
        tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
        _cabi.check(L.gdn_pr_plan_kernel_time(plan, 0, 0, tot, C.byref(n)))
        a, b = tot[0] / n.value, tot[1] / n.value
        res[v].append((a, b))
        if rnd == 0:
            _cabi.check(L.gdn_pr_export_dev(plan, state, scores, 0.85, None))
            _cabi.check(L.gdn_dev_download(out.ctypes.data_as(C.c_void_p), scores, 4 * m))
            crc[v] = zlib.crc32(out.tobytes())
        print("round %d %s=%s: A %.3f  B %.3f  sum %.3f ms" % (rnd, name, v, a, b, a + b), flush=True)
        L.gdn_pr_plan_free(plan)
        for p in (state, c0, c1, diff):
            L.gdn_dev_free(p)
for v in (va, vb):
    arr = np.array(res[v])
    print("%s=%s: A median %.3f  B median %.3f  sum median %.3f  min %.3f  (crc %08x)" % (
        name, v, np.median(arr[:, 0]), np.median(arr[:, 1]), np.median(arr.sum(1)), arr.sum(1).min(), crc.get(v, 0)))
print("same bits:", crc.get(va) == crc.get(vb))
