#!/usr/bin/env python3
"""A/B of an SSSP PLAN-BUILD knob: python tools/sssp_ab_plan.py NAME VAL_A VAL_B [scale] [rounds]
One process, one graph (R-MAT, U[1,255] weights and unit weights); every round builds a plan under each value and solves
6 times from the bench's source; distances must agree."""
import ctypes as C
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
name, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
scale = int(sys.argv[4]) if len(sys.argv) > 4 else 24
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda", 0)
go = C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None))
m, nnz = m.value, nnz.value
torch.manual_seed(5)
p = lambda t: C.c_void_p(t.data_ptr())
deg = torch.empty(m, dtype=torch.int32, device=dev)
_cabi.check(L.gdn_graph_degrees_dev(go, p(deg), None))
src = int(torch.nonzero(deg[:1 << 16] > 0)[0].item())
dist = torch.empty(m, dtype=torch.int32, device=dev)
for label, w in (("U[1,255]", torch.randint(1, 256, (nnz,), dtype=torch.int32, device=dev)), ("unit", torch.ones(nnz, dtype=torch.int32, device=dev))):
    res, crc = {va: [], vb: []}, {}
    for rnd in range(rounds):
        for v in (va, vb) if rnd % 2 == 0 else (vb, va):
            _cabi.check(L.gdn_option_set(name.encode(), v.encode()))
            plan = C.c_void_p()
            _cabi.check(L.gdn_sssp_plan_create(go, p(w), 1, C.byref(plan)))
            ts = []
            for _ in range(6):
                st = _cabi.GdnStats()
                _cabi.check(L.gdn_sssp_run(plan, src, 16, p(dist), C.byref(st)))
                ts.append(st.solve_ms)
            res[v].append(float(np.median(ts)))
            crc[v] = zlib.crc32(dist.cpu().numpy().tobytes())
            print("%s round %d %s=%s: %s" % (label, rnd, name, v, " ".join("%.3f" % t for t in ts)), flush=True)
            L.gdn_sssp_plan_free(plan)
    for v in (va, vb):
        print("%s %s=%s: median of plan medians %.3f ms, min %.3f (crc %08x)" % (label, name, v, np.median(res[v]), min(res[v]), crc[v]))
    print(label, "same distances:", crc[va] == crc[vb])
