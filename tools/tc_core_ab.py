#!/usr/bin/env python3
"""The forward triangle count with cores of 0 / 4096 / 8192 / 16384 ranks on ONE graph in one process (symmetrized R-MAT
scale S, oriented on the device): plan build, count (median / min of `reps`), the same total.   tc_core_ab.py S [reps]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 23
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
go, gs, dag = C.c_void_p(), C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
_cabi.check(L.gdn_graph_symmetrize(go, C.byref(gs)))
L.gdn_graph_free(go)
_cabi.check(L.gdn_graph_orient(gs, C.byref(dag)))
L.gdn_graph_free(gs)
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(dag, C.byref(m), C.byref(nnz), None, None))
_cabi.check(L.gdn_option_set(b"GDN_TC_FORM", b"f"))
totals = set()
for rnd in range(2):
    for k in [int(x) for x in os.environ.get("TC_AB_CORES", "0,4096,8192,12288,16384").split(",")]:
        _cabi.check(L.gdn_option_set(b"GDN_TC_CORE", str(k).encode()))
        plan = C.c_void_p()
        t0 = time.time()
        _cabi.check(L.gdn_tc_plan_create(dag, 1, C.byref(plan)))
        tb = time.time() - t0
        ms = []
        total = C.c_uint64(0)
        for i in range(reps + 1):
            st = _cabi.GdnStats()
            _cabi.check(L.gdn_tc_plan_count(plan, C.byref(total), C.byref(st)))
            if i:
                ms.append(st.solve_ms)
        L.gdn_tc_plan_free(plan)
        ms.sort()
        totals.add(total.value)
        print("RMAT-%d dag %d core %5d (reported %5d): plan %.3f s  count median %.3f min %.3f ms  %.2f G dag edges/s  triangles %d" % (
            scale, nnz.value, k, st.reserved >> 8, tb, ms[len(ms) // 2], ms[0], nnz.value / ms[len(ms) // 2] / 1e6, total.value), flush=True)
print("same total:", len(totals) == 1)
