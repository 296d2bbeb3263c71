#!/usr/bin/env python3
"""A/B of triangle-count knobs on ONE resident DAG (symmetrized R-MAT scale S, oriented on the device; the plan is rebuilt under
every knob set): count median / min of `reps`, the same total.  tc_knob_ab.py <S | orkut> reps "K1=V1,K2=V2" "K1=V3" ...  ("" = defaults)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
arg = sys.argv[1] if len(sys.argv) > 1 else "23"  # an R-MAT scale, or "orkut" = the Orkut-like stand-in of graphio.ORKUT_LIKE
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sets = sys.argv[3:] or [""]
go, gs, dag = C.c_void_p(), C.c_void_p(), C.c_void_p()
if arg == "orkut":
    r = graphio.ORKUT_LIKE
    scale = r["scale"]
    _cabi.check(L.gdn_rmat_build_ex(r["scale"], r["n_edges"], *r["abc"], graphio.K_RAND_SEED, r["flags"], C.byref(go), None))
else:
    scale = int(arg)
    _cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
_cabi.check(L.gdn_graph_symmetrize(go, C.byref(gs)))
L.gdn_graph_free(go)
_cabi.check(L.gdn_graph_orient(gs, C.byref(dag)))
L.gdn_graph_free(gs)
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(dag, C.byref(m), C.byref(nnz), None, None))
totals = set()
for rnd in range(2):
    for spec in sets:
        env = dict(kv.split("=") for kv in spec.split(",") if kv)
        for k, v in env.items():
            _cabi.check(L.gdn_option_set(k.encode(), v.encode()))
        plan = C.c_void_p()
        _cabi.check(L.gdn_tc_plan_create(dag, 1, C.byref(plan)))
        ms, total = [], C.c_uint64(0)
        for i in range(reps + 1):
            st = _cabi.GdnStats()
            _cabi.check(L.gdn_tc_plan_count(plan, C.byref(total), C.byref(st)))
            if i:
                ms.append(st.solve_ms)
        L.gdn_tc_plan_free(plan)
        for k in env:
            _cabi.check(L.gdn_option_set(k.encode(), None))
        ms.sort()
        totals.add(total.value)
        print("%s-%d dag %d [%-40s] core %5d: count median %.3f min %.3f ms  %.2f G dag edges/s  triangles %d" % (
            "orkut-like" if arg == "orkut" else "RMAT", scale, nnz.value, spec, st.reserved >> 8, ms[len(ms) // 2], ms[0], nnz.value / ms[len(ms) // 2] / 1e6, total.value), flush=True)
print("same total:", len(totals) == 1)
