#!/usr/bin/env python3
"""SpMV timing through the C-ABI only (no torch in the process): PB layout on RMAT-<scale> x16 with Ax, x ~ U(0,1);
prints the per-phase kernel times and a CRC of y (the layouts / tiers must leave it unchanged)."""
import ctypes as C
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 25
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
y = dev(np.zeros(m, np.float32))
plan = C.c_void_p()
_cabi.check(L.gdn_spmv_plan_create(gi, Ax, 1, C.byref(plan)))
_cabi.check(L.gdn_spmv_dev(plan, Ax, x, y, None))
out = np.empty(m, np.float32)
_cabi.check(L.gdn_dev_download(out.ctypes.data_as(C.c_void_p), y, 4 * m))
print("check: sum %.6f crc %08x" % (float(out.astype(np.float64).sum()), zlib.crc32(out.tobytes())))
steps, best = 10, None
for batch in range(3):
    _cabi.check(L.gdn_spmv_plan_kernel_time(plan, 1, steps, None, None))
    for _ in range(steps):
        _cabi.check(L.gdn_spmv_dev(plan, Ax, x, y, None))
    tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
    _cabi.check(L.gdn_spmv_plan_kernel_time(plan, 0, 0, tot, C.byref(n)))
    cur = (tot[0] / n.value, tot[1] / n.value)
    if best is None or sum(cur) < sum(best):
        best = cur
_cabi.check(L.gdn_spmv_plan_check(plan))
b = L.gdn_spmv_bytes(plan)
print("spmv scale %d nnz %d: A %.3f ms  B %.3f ms  sum %.3f ms  = %.0f GB/s algorithmic (%.1f %% of 8 TB/s)" % (
    scale, nnz, best[0], best[1], sum(best), b / sum(best) / 1e6, 100.0 * b / sum(best) / 1e6 / 8000.0))
