#!/usr/bin/env python3
"""Betweenness centrality (one source) timing through the C-ABI only: resident R-MAT graph of scale S."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
go = C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None))
m = m.value
deg = C.c_void_p()
_cabi.check(L.gdn_dev_alloc(4 * m, C.byref(deg)))
_cabi.check(L.gdn_graph_degrees_dev(go, deg, None))
hdeg = np.empty(1 << 16, np.int32)
_cabi.check(L.gdn_dev_download(hdeg.ctypes.data_as(C.c_void_p), deg, 4 * (1 << 16)))
sources = np.nonzero(hdeg > 0)[0][:3].tolist()
scores = C.c_void_p()
_cabi.check(L.gdn_dev_alloc(4 * m, C.byref(scores)))
zero = np.zeros(m, np.float32)
use_plan = len(sys.argv) > 2 and sys.argv[2] == "plan"
plan = C.c_void_p()
if use_plan:
    _cabi.check(L.gdn_bc_plan_create(go, None, C.byref(plan)))
out = np.empty(m, np.float32)
for s in sources + sources[:1]:
    _cabi.check(L.gdn_dev_upload(scores, zero.ctypes.data_as(C.c_void_p), 4 * m))
    st = _cabi.GdnStats()
    if use_plan:
        _cabi.check(L.gdn_bc_run(plan, int(s), scores, C.byref(st)))
    else:
        _cabi.check(L.gdn_bc_dev(go, int(s), scores, C.byref(st)))
    _cabi.check(L.gdn_dev_download(out.ctypes.data_as(C.c_void_p), scores, 4 * m))
    print("BC%s RMAT-%d from %d: %.3f ms, %d levels, %d edge visits, %.1f GTEPS (prep %.0f ms; sum of scores %.6f)" % (
        " plan" if use_plan else "", scale, s, st.solve_ms, st.iterations, st.edges_traversed,
        st.edges_traversed / st.solve_ms / 1e6, st.prep_ms, float(out.astype(np.float64).sum())), flush=True)
