#!/usr/bin/env python3
"""Where a BFS on a hub-less graph spends its time: per-level trace (GDN_BFS_TRACE) of the resident plan on
the uniform random and the small-world graph of tools/shapes.py `large`, under the default engine choice and with the heavy
levels forced onto each engine.  usage: bfs_shapes_trace.py [uniform|small_world|rmat] ..."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
SHAPES = {"uniform": lambda: graphio.uniform_edges(1 << 23, 1 << 26, 7), "small_world": lambda: graphio.small_world_edges(1 << 22, 16, 0.1, 7)}
KNOBS = [("default", {}),
         ("heavy levels = dense sweeps", {"GDN_BFS_BU_FRAC": "0", "GDN_BFS_BU_EDGE_DIV": "0", "GDN_BFS_BTD": "0"}),
         ("no early bottom-up (share rule off)", {"GDN_BFS_BU_EDGE_DIV": "0"}),
         ("no binned level", {"GDN_BFS_BTD": "0"}),
         ("bottom-up from 1/8 of the edges", {"GDN_BFS_BU_EDGE_DIV": "8"}),
         ("bottom-up from 1/5 of the edges", {"GDN_BFS_BU_EDGE_DIV": "5"}),
         ("bottom-up from 1/4 of the edges", {"GDN_BFS_BU_EDGE_DIV": "4"}),
         ("dense from nnz/64", {"GDN_BFS_ALPHA_DENSE": "64"}),
         ("dense from nnz/128, heavy = dense", {"GDN_BFS_ALPHA_DENSE": "128", "GDN_BFS_BU_FRAC": "0", "GDN_BFS_BU_EDGE_DIV": "0", "GDN_BFS_BTD": "0"})]
for name in (sys.argv[1:] or ["uniform", "small_world"]):
    ho, hi = C.c_void_p(), C.c_void_p()
    if name == "rmat":
        _cabi.check(L.gdn_rmat_build(24, 16, graphio.K_RAND_SEED, 1, C.byref(ho), C.byref(hi)))
    elif name.startswith("uniform") and name != "uniform":  # uniform<scale>: 2^scale vertices, 16 x 2^scale draws, generated on the device
        sc = int(name[7:])
        _cabi.check(L.gdn_rmat_build_ex(sc, 16 << sc, 0.25, 0.25, 0.25, graphio.K_RAND_SEED, 1, C.byref(ho), C.byref(hi)))
    else:
        m, src, dst = SHAPES[name]()
        g = graphio.build_csr_device(m, src, dst)
        _cabi.check(L.gdn_graph_upload(g.m, g.nnz, g.rowptr.ctypes.data_as(C.c_void_p), g.colidx.ctypes.data_as(C.c_void_p), C.byref(ho)))
        _cabi.check(L.gdn_graph_transpose(ho, C.byref(hi)))
    mm, nn = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(ho, C.byref(mm), C.byref(nn), None, None))
    deg = np.empty(mm.value, np.int32)
    d_deg = C.c_void_p()
    _cabi.check(L.gdn_dev_alloc(4 * mm.value, C.byref(d_deg)))
    _cabi.check(L.gdn_graph_degrees_dev(ho, d_deg, None))
    _cabi.check(L.gdn_dev_download(deg.ctypes.data_as(C.c_void_p), d_deg, 4 * mm.value))
    s = int(np.nonzero(deg > 0)[0][0])
    d_dist = C.c_void_p()
    _cabi.check(L.gdn_dev_alloc(4 * mm.value, C.byref(d_dist)))
    plan = C.c_void_p()
    _cabi.check(L.gdn_bfs_plan_create(ho, hi, 1, C.byref(plan)))
    ref = None
    for label, env in KNOBS:
        for k, v in env.items():
            _cabi.check(L.gdn_option_set(k.encode(), v.encode()))
        best = None
        for rep in range(4):
            st = _cabi.GdnStats()
            if rep == 3:
                _cabi.check(L.gdn_option_set(b"GDN_BFS_TRACE", b"1"))
                print("== %s (|V| %d |E| %d, source %d): %s" % (name, mm.value, nn.value, s, label), file=sys.stderr, flush=True)
            _cabi.check(L.gdn_bfs_run(plan, s, d_dist, C.byref(st)))
            if rep < 3:
                best = st.solve_ms if best is None else min(best, st.solve_ms)
        _cabi.check(L.gdn_option_set(b"GDN_BFS_TRACE", None))
        got = np.empty(mm.value, np.int32)
        _cabi.check(L.gdn_dev_download(got.ctypes.data_as(C.c_void_p), d_dist, 4 * mm.value))
        if ref is None:
            ref = got
        assert np.array_equal(got, ref), label
        print("%-12s %-40s %.3f ms  %d levels  %.1f GTEPS" % (name, label, best, st.iterations, st.edges_traversed / best / 1e6), flush=True)
        for k in env:
            _cabi.check(L.gdn_option_set(k.encode(), None))
    L.gdn_bfs_plan_free(plan)
    L.gdn_graph_free(ho)
    L.gdn_graph_free(hi)
