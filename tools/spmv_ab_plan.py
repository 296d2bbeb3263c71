#!/usr/bin/env python3
"""A/B of an SpMV PLAN-BUILD knob: python tools/spmv_ab_plan.py NAME VAL_A VAL_B [scale] [rounds]  (one process, one matrix;
every round builds a plan under each value, 10 multiplies timed with the plan's events; y must have the same bits)"""
import ctypes as C
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
name, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
scale = int(sys.argv[4]) if len(sys.argv) > 4 else 25
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
_cabi.check(L.gdn_option_set(b"GDN_SPMV_PLACE", b"0"))
gi = C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, None, C.byref(gi)))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(gi, C.byref(m), C.byref(nnz), None, None))
m, nnz = m.value, nnz.value


def dev(a):
    p = C.c_void_p()
    _cabi.check(L.gdn_dev_alloc(a.nbytes, C.byref(p)))
    _cabi.check(L.gdn_dev_upload(p, a.ctypes.data_as(C.c_void_p), a.nbytes))
    return p


rng = np.random.default_rng(7)
Ax = dev(rng.random(nnz, dtype=np.float32))
x = dev(rng.random(m, dtype=np.float32))
zeros = np.zeros(m, np.float32)
y = dev(zeros)
out = np.empty(m, np.float32)
res, crc = {va: [], vb: []}, {}
for rnd in range(rounds):
    for v in (va, vb) if rnd % 2 == 0 else (vb, va):
        for nm in name.split(","):
            _cabi.check(L.gdn_option_set(nm.encode(), v.encode()))
        plan = C.c_void_p()
        _cabi.check(L.gdn_spmv_plan_create(gi, Ax, 1, C.byref(plan)))
        _cabi.check(L.gdn_dev_upload(y, zeros.ctypes.data_as(C.c_void_p), 4 * m))
        _cabi.check(L.gdn_spmv_dev(plan, Ax, x, y, None))
        _cabi.check(L.gdn_dev_download(out.ctypes.data_as(C.c_void_p), y, 4 * m))
        crc[v] = zlib.crc32(out.tobytes())
        steps = 10
        _cabi.check(L.gdn_spmv_plan_kernel_time(plan, 1, steps, None, None))
        for _ in range(steps):
            _cabi.check(L.gdn_spmv_dev(plan, Ax, x, y, None))
        tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
        _cabi.check(L.gdn_spmv_plan_kernel_time(plan, 0, 0, tot, C.byref(n)))
        a, b = tot[0] / n.value, tot[1] / n.value
        res[v].append((a, b))
        print("round %d %s=%s: A %.3f  B %.3f  sum %.3f ms" % (rnd, name, v, a, b, a + b), flush=True)
        _cabi.check(L.gdn_spmv_plan_check(plan))
        L.gdn_spmv_plan_free(plan)
for v in (va, vb):
    arr = np.array(res[v])
    print("%s=%s: A median %.3f  B median %.3f  sum median %.3f  min %.3f  (crc %08x)" % (
        name, v, np.median(arr[:, 0]), np.median(arr[:, 1]), np.median(arr.sum(1)), arr.sum(1).min(), crc[v]))
print("same bits:", crc[va] == crc[vb])
