#!/usr/bin/env python3
"""PageRank iteration time of the blocked layout on a NON-R-MAT shape under plan-build knobs, one process, one graph:
   python3 tools/pr_shape_knobs.py uniform|small_world "GDN_PB_SLICES_LOG=10" "GDN_PB_LOG_CHUNK=14 GDN_PB_LOG_BIN=13" ...
Every argument after the shape is one configuration (space-separated NAME=VALUE pairs; "" = the defaults).  Prints the
kernel time of an iteration (HIP events on the launch stream, 10 iterations), the roofline fraction on the plan's own byte
model and a CRC of the scores after 13 iterations (equal bits across configurations up to the summation order of a bin)."""
import ctypes as C
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
shape = sys.argv[1] if len(sys.argv) > 1 else "uniform"
configs = sys.argv[2:] or [""]
make = {"uniform": lambda: graphio.uniform_edges(1 << 23, 1 << 26, 7),
        "small_world": lambda: graphio.small_world_edges(1 << 22, 16, 0.1, 7)}[shape]
p = lambda a: a.ctypes.data_as(C.c_void_p)
m, src, dst = make()
g = graphio.build_csr_device(m, src, dst)
del src, dst
ho, hi = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_graph_upload(g.m, g.nnz, p(g.rowptr), p(g.colidx), C.byref(ho)))
_cabi.check(L.gdn_graph_transpose(ho, C.byref(hi)))
deg = g.degrees().astype(np.int32)
d_deg, d_s, d_c0, d_c1, d_diff = (C.c_void_p() for _ in range(5))
for d, n in ((d_deg, 4 * g.m), (d_s, 4 * g.m), (d_c0, 4 * g.m + 16), (d_c1, 4 * g.m + 16), (d_diff, 8)):
    _cabi.check(L.gdn_dev_alloc(n, C.byref(d)))
_cabi.check(L.gdn_dev_upload(d_deg, p(deg), 4 * g.m))
print("%s: |V| %d |E| %d" % (shape, g.m, g.nnz), flush=True)
for cfg in configs:
    pairs = [kv.split("=", 1) for kv in cfg.split()]
    for k, v in pairs:
        _cabi.check(L.gdn_option_set(k.encode(), v.encode()))
    for layout, lname in ((_cabi.GDN_LAYOUT_PB, "pb"), (_cabi.GDN_LAYOUT_PB_SQUISHED, "pb squished")):
        plan = C.c_void_p()
        _cabi.check(L.gdn_pr_plan_create(hi, d_deg, g.m, 0, layout, C.byref(plan)))
        init = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        _cabi.check(L.gdn_dev_upload(d_s, p(init), 4 * g.m))
        _cabi.check(L.gdn_pr_contrib_dev(plan, d_s, d_c0, None))
        cin, cout = d_c0, d_c1
        for _ in range(3):
            _cabi.check(L.gdn_pr_pull_dev(plan, cin, d_s, cout, d_diff, 0.85, None))
            cin, cout = cout, cin
        _cabi.check(L.gdn_pr_plan_kernel_time(plan, 1, 10, None, None))
        for _ in range(10):
            _cabi.check(L.gdn_pr_pull_dev(plan, cin, d_s, cout, d_diff, 0.85, None))
            cin, cout = cout, cin
        tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
        _cabi.check(L.gdn_pr_plan_kernel_time(plan, 0, 0, tot, C.byref(n)))
        k_ms = (tot[0] + tot[1]) / max(n.value, 1)
        b = int(L.gdn_pr_iter_bytes(plan))
        sc = np.empty(g.m, np.float32)
        if hasattr(L, "gdn_pr_plan_export_scores"):
            pass
        _cabi.check(L.gdn_dev_download(p(sc), d_s, 4 * g.m))
        print("  %-44s %-12s A %.4f + B %.4f = %.4f ms  frac %.3f  crc %08x" % (
            cfg or "(defaults)", lname, tot[0] / max(n.value, 1), tot[1] / max(n.value, 1), k_ms, b / (k_ms * 1e-3) / 1e9 / 8000.0,
            zlib.crc32(sc.tobytes())), flush=True)
        L.gdn_pr_plan_free(plan)
    for k, v in pairs:
        _cabi.check(L.gdn_option_set(k.encode(), None))
