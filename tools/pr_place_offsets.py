#!/usr/bin/env python3
"""Does the speed of a placement of PageRank's `vals` belong to the ALLOCATION or to the ADDRESS?  One RMAT-27 plan whose placement
search times every candidate at offset 0 and at seven offsets inside the same allocation (GDN_PLACE_OFFSETS=1; candidates are
256 MB longer), every candidate timed (GDN_PR_PLACE_STOP=0), trace on stderr.  usage: pr_place_offsets.py [scale] [candidates] [ab]   (ab: phase A and phase B of every candidate instead of the offsets)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
cands = sys.argv[2] if len(sys.argv) > 2 else "10"
knobs = [("GDN_PR_PLACE_TRACE", "1"), ("GDN_PR_PLACE_STOP", "0"), ("GDN_PR_PLACE_VALS", cands), ("GDN_PR_PLACE_BUDGET_MS", "60000")]
if len(sys.argv) > 3 and sys.argv[3] == "ab":  # both phases of every candidate instead of the offsets inside it
    knobs.append(("GDN_PR_PLACE_TRACE_AB", "1"))
else:
    knobs.append(("GDN_PLACE_OFFSETS", "1"))
for k, v in knobs:
    _cabi.check(L.gdn_option_set(k.encode(), v.encode()))
go, gi = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(gi, C.byref(m), C.byref(nnz), None, None))
deg = C.c_void_p()
_cabi.check(L.gdn_dev_alloc(4 * m.value, C.byref(deg)))
_cabi.check(L.gdn_graph_degrees_dev(go, deg, None))
L.gdn_graph_free(go)
plan = C.c_void_p()
_cabi.check(L.gdn_pr_plan_create(gi, deg, m.value, 0, 2, C.byref(plan)))
print("plan created: %d vertices, %d edges" % (m.value, nnz.value))
L.gdn_pr_plan_free(plan)
