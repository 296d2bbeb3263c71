#!/usr/bin/env python3
"""Delta PageRank (src/pr/delta.cu semantics) timing through the C-ABI only: resident R-MAT graph of scale S, the
plain pull PageRank of the same graph to the same stop (L1 change < 1e-4) beside it."""
import ctypes as C
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
push_div = int(sys.argv[2]) if len(sys.argv) > 2 else 8
go, gi = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
_cabi.check(L.gdn_graph_transpose(go, C.byref(gi)))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None))
m, nnz = m.value, nnz.value
t0 = time.time()
plan = C.c_void_p()
_cabi.check(L.gdn_pr_delta_plan_create(gi, go, _cabi.GDN_LAYOUT_AUTO, C.byref(plan)))
print("RMAT-%d: m %d nnz %d, delta plan built in %.2f s" % (scale, m, nnz, time.time() - t0), flush=True)
scores = C.c_void_p()
_cabi.check(L.gdn_dev_alloc(4 * m, C.byref(scores)))
init = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
out = np.empty(m, np.float32)
for rep in range(3):
    _cabi.check(L.gdn_dev_upload(scores, init.ctypes.data_as(C.c_void_p), 4 * m))
    st = _cabi.GdnStats()
    _cabi.check(L.gdn_pr_delta_run(plan, scores, 0.85, 1e-4, 1e-3, 100, push_div, C.byref(st)))
    n = C.c_int32()
    diff, items, mode = np.zeros(100), np.zeros(100, np.int32), np.zeros(100, np.int32)
    _cabi.check(L.gdn_pr_delta_trace(plan, 100, C.byref(n), diff.ctypes.data_as(C.c_void_p),
                                     items.ctypes.data_as(C.c_void_p), mode.ctypes.data_as(C.c_void_p)))
    _cabi.check(L.gdn_dev_download(out.ctypes.data_as(C.c_void_p), scores, 4 * m))
    k = n.value
    print("delta PR RMAT-%d: %.3f ms, %d iterations (%d pull, %d push as masked pull, %d push with atomics), %.3f ms/iteration, last L1 %.3e, crc %08x" % (
        scale, st.solve_ms, st.iterations, int((mode[:k] == 0).sum()), int((mode[:k] == 3).sum()), int((mode[:k] == 1).sum()), st.solve_ms / st.iterations,
        st.last_error, zlib.crc32(out.tobytes())), flush=True)
print("  frontier after each iteration:", items[:k].tolist())
print("  L1 norm of the deltas:", ["%.2e" % d for d in diff[:k]])
L.gdn_pr_delta_plan_free(plan)

# plain pull PageRank (gdn_pr_plan_*) to the reference's stop on the same graph
deg = C.c_void_p()
_cabi.check(L.gdn_dev_alloc(4 * m, C.byref(deg)))
_cabi.check(L.gdn_graph_degrees_dev(go, deg, None))
pl = C.c_void_p()
_cabi.check(L.gdn_pr_plan_create(gi, deg, m, 0, _cabi.GDN_LAYOUT_PB_SQUISHED if nnz >= (1 << 22) else _cabi.GDN_LAYOUT_AUTO,
                                 C.byref(pl)))
ms = C.c_int32()
_cabi.check(L.gdn_pr_plan_state_size(pl, C.byref(ms)))
state, c0, c1, dd = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
for b in (state, c0, c1):
    _cabi.check(L.gdn_dev_alloc(4 * ms.value, C.byref(b)))
_cabi.check(L.gdn_dev_alloc(8, C.byref(dd)))
for rep in range(2):
    _cabi.check(L.gdn_dev_upload(scores, init.ctypes.data_as(C.c_void_p), 4 * m))
    t0 = time.time()
    _cabi.check(L.gdn_pr_import_dev(pl, scores, state, 0.85, None))
    dead = C.c_double()
    _cabi.check(L.gdn_pr_import_diff(pl, C.byref(dead)))
    _cabi.check(L.gdn_pr_contrib_dev(pl, state, c0, None))
    cin, cout = c0, c1
    hd = np.zeros(1)
    for it in range(100):
        _cabi.check(L.gdn_pr_pull_dev(pl, cin, state, cout, dd, 0.85, None))
        _cabi.check(L.gdn_dev_download(hd.ctypes.data_as(C.c_void_p), dd, 8))
        d = hd[0] + (dead.value if it == 0 else 0.0)
        cin, cout = cout, cin
        if d < 1e-4:
            break
    _cabi.check(L.gdn_pr_export_dev(pl, state, scores, 0.85, None))
    _cabi.check(L.gdn_dev_download(out.ctypes.data_as(C.c_void_p), scores, 4 * m))
    el = (time.time() - t0) * 1e3
    print("plain pull PR RMAT-%d: %d iterations to L1 %.3e, %.3f ms wall incl. the score download" % (scale, it + 1, d, el))
