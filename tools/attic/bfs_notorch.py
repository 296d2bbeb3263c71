#!/usr/bin/env python3
"""BFS timing through the C-ABI only (no torch): resident plan on R-MAT scale S; GDN_BFS_TRACE=1 prints per-level times."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
go, gi = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None))
m = m.value
deg = C.c_void_p()
_cabi.check(L.gdn_dev_alloc(4 * m, C.byref(deg)))
_cabi.check(L.gdn_graph_degrees_dev(go, deg, None))
hdeg = np.empty(1 << 16, np.int32)
_cabi.check(L.gdn_dev_download(hdeg.ctypes.data_as(C.c_void_p), deg, 4 * (1 << 16)))
sources = np.nonzero(hdeg > 0)[0][:3].tolist()
dist = C.c_void_p()
_cabi.check(L.gdn_dev_alloc(4 * m, C.byref(dist)))
plan = C.c_void_p()
_cabi.check(L.gdn_bfs_plan_create(go, gi, 1, C.byref(plan)))
for s in sources + sources[:1]:
    st = _cabi.GdnStats()
    _cabi.check(L.gdn_bfs_run(plan, int(s), dist, C.byref(st)))
    print("BFS RMAT-%d from %d: %.3f ms, %d levels, %d edges, %.1f GTEPS" % (
        scale, s, st.solve_ms, st.iterations, st.edges_traversed, st.edges_traversed / st.solve_ms / 1e6), flush=True)
