#!/usr/bin/env python3
"""The searches of bench.py's BFS block IN ORDER (one run of the slow source in four jumps from 2.0 to 3.0 ms --
which one, and which level?): R-MAT scale S, the first three non-isolated sources, <reps> consecutive searches each, solve_ms of
every search in issue order; then one traced search per source (GDN_BFS_TRACE: per-level engine, frontier, time).
usage: bfs_runs.py [scale 27] [reps 6] [rounds 2]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
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
for rnd in range(rounds):
    for s in sources:
        ms = []
        for _ in range(reps):
            st = _cabi.GdnStats()
            _cabi.check(L.gdn_bfs_run(plan, int(s), dist, C.byref(st)))
            ms.append(st.solve_ms)
        print("round %d source %d: %s ms (levels %d, edges %d)" % (rnd, s, " ".join("%.3f" % x for x in ms), st.iterations, st.edges_traversed), flush=True)
_cabi.check(L.gdn_option_set(b"GDN_BFS_TRACE", b"1"))
for s in sources:
    print("---- traced search from %d" % s, flush=True)
    sys.stdout.flush()
    st = _cabi.GdnStats()
    _cabi.check(L.gdn_bfs_run(plan, int(s), dist, C.byref(st)))
    sys.stderr.flush()
    print("traced: %.3f ms" % st.solve_ms, flush=True)
