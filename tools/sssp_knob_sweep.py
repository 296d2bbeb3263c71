#!/usr/bin/env python3
"""Run-time knobs of the SSSP schedule on ONE resident plan (R-MAT scale S, weights U[1,255], delta 16): per knob set the
median of 8 solves, the phase count, distances compared with the first set's.
usage: sssp_knob_sweep.py <scale> "K=V,K=V" ...   ("" = defaults)"""
import ctypes as C
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sets = sys.argv[2:] or [""]
dev = torch.device("cuda", 0)
go = C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None))
m, nnz = m.value, nnz.value
gen = torch.Generator(device=dev)
gen.manual_seed(5)
p = lambda t: C.c_void_p(t.data_ptr())
deg = torch.empty(m, dtype=torch.int32, device=dev)
_cabi.check(L.gdn_graph_degrees_dev(go, p(deg), None))
src = int(torch.nonzero(deg[:1 << 16] > 0)[0].item())
_ = torch.ones(nnz, dtype=torch.int32, device=dev)  # (bench.py draws the unit weights first: the same generator state)
w = torch.randint(1, 256, (nnz,), dtype=torch.int32, device=dev, generator=gen)
dist = torch.empty(m, dtype=torch.int32, device=dev)
plan = C.c_void_p()
_cabi.check(L.gdn_sssp_plan_create(go, p(w), 1, C.byref(plan)))
ref = None
for spec in sets:
    env = dict(kv.split("=") for kv in spec.split(",") if kv)
    delta = int(env.pop("delta", 16))
    for k, v in env.items():
        _cabi.check(L.gdn_option_set(k.encode(), v.encode()))
    ts = []
    for _ in range(9):
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_sssp_run(plan, src, delta, p(dist), C.byref(st)))
        ts.append(st.solve_ms)
    crc = zlib.crc32(dist.cpu().numpy().tobytes())
    ref = crc if ref is None else ref
    print("%-44s median %.3f min %.3f ms  %d phases  relaxed %.2fx%s" % ("[" + spec + "]", float(np.median(ts[1:])), min(ts[1:]), st.iterations,
                                                                   st.last_error / max(st.edges_traversed, 1), "" if crc == ref else "  DISTANCES DIFFER"), flush=True)
    for k in env:
        _cabi.check(L.gdn_option_set(k.encode(), None))
