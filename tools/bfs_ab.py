#!/usr/bin/env python3
"""A/B of BFS knobs on ONE resident graph (R-MAT scale S; the plan is rebuilt under every knob set, so plan-build knobs such as
GDN_BFS_HUBS2 count too): per knob set the three bench sources, best of 3 untraced runs + one traced run (GDN_BFS_TRACE),
depths compared with the first set's.  usage: bfs_ab.py <scale | u<scale>> "K1=V1,K2=V2" "K1=V3" ...  ("" = defaults; u26 = uniform random 2^26 x 16)"""
import ctypes as C
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
arg = sys.argv[1] if len(sys.argv) > 1 else "27"  # "27": R-MAT scale 27; "u26": uniform random, 2^26 vertices x 16 draws each
scale = int(arg.lstrip("u"))
sets = sys.argv[2:] or [""]
go, gi = C.c_void_p(), C.c_void_p()
if arg.startswith("u"):
    _cabi.check(L.gdn_rmat_build_ex(scale, 16 << scale, 0.25, 0.25, 0.25, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
else:
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
ref = {}
for spec in sets:
    env = dict(kv.split("=") for kv in spec.split(",") if kv)
    for k, v in env.items():
        _cabi.check(L.gdn_option_set(k.encode(), v.encode()))
    plan = C.c_void_p()
    _cabi.check(L.gdn_bfs_plan_create(go, gi, 1, C.byref(plan)))
    line = []
    for s in sources:
        best = None
        for rep in range(4):
            if rep == 3:
                _cabi.check(L.gdn_option_set(b"GDN_BFS_TRACE", b"1"))
                print("== [%s] source %d" % (spec, s), file=sys.stderr, flush=True)
            st = _cabi.GdnStats()
            _cabi.check(L.gdn_bfs_run(plan, int(s), dist, C.byref(st)))
            if rep < 3:
                best = st.solve_ms if best is None else min(best, st.solve_ms)
        _cabi.check(L.gdn_option_set(b"GDN_BFS_TRACE", None))
        h = np.empty(m, np.int32)
        _cabi.check(L.gdn_dev_download(h.ctypes.data_as(C.c_void_p), dist, 4 * m))
        crc = zlib.crc32(h.tobytes())
        ref.setdefault(s, crc)
        line.append("%d: %.3f ms%s" % (s, best, "" if crc == ref[s] else " DEPTHS DIFFER"))
    print("%-50s %s" % ("[" + spec + "]", "   ".join(line)), flush=True)
    L.gdn_bfs_plan_free(plan)
    for k in env:
        _cabi.check(L.gdn_option_set(k.encode(), None))
