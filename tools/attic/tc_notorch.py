#!/usr/bin/env python3
"""Triangle counting timing through the C-ABI only (no torch): R-MAT scale S, symmetrized and oriented on the device."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 21
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
go, gs = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
_cabi.check(L.gdn_graph_symmetrize(go, C.byref(gs)))
L.gdn_graph_free(go)
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(gs, C.byref(m), C.byref(nnz), None, None))
for r in range(reps):
    total = C.c_uint64(0)
    st = _cabi.GdnStats()
    _cabi.check(L.gdn_tc_dev(gs, 0, C.byref(total), C.byref(st)))
    print("RMAT-%d sym: |V| %d |E| %d dag %d triangles %d count %.3f ms orient %.3f ms  %.3f G dag edges/s" % (
        scale, m.value, nnz.value, st.edges_traversed, total.value, st.solve_ms, st.prep_ms,
        st.edges_traversed / st.solve_ms / 1e6))
# the row-range shards of a multi-GPU count, one after the other on this device (balance check): tc_notorch.py S reps N
if len(sys.argv) > 3:
    from gardenia_amd.sharded import HipTCBackend, ShardedTC
    world = int(sys.argv[3])
    be = HipTCBackend(gs, oriented=False, device=None)
    rp = be.rowptr()
    ms, cnt = [], []
    for r in range(world):
        p = ShardedTC(be, rp, r, world, dist=None)
        best = 1e30
        for _ in range(2):
            c = be.count_rows(p.lo, p.hi)
            best = min(best, be.last_ms)
        ms.append(best)
        cnt.append(c)
    print("%d row-range shards of equal DAG-edge count: sum %d, ms per shard %s, max/mean %.2f" % (
        world, sum(cnt), ["%.2f" % t for t in ms], max(ms) / (sum(ms) / world)))
    be.close()
