"""Placement spread of the SSSP plan: K plans alive in one process, each solved several times round-robin.
usage: sssp_replan.py [scale] [K]"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from gardenia_amd import _cabi, graphio
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = _cabi.lib(); dev = torch.device("cuda", 0)
go = C.c_void_p(); _cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
m, nnz = C.c_int32(), C.c_uint64(); _cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None)); m, nnz = m.value, nnz.value
torch.manual_seed(5)
w = torch.randint(1, 256, (nnz,), dtype=torch.int32, device=dev)
deg = torch.empty(m, dtype=torch.int32, device=dev); _cabi.check(L.gdn_graph_degrees_dev(go, C.c_void_p(deg.data_ptr()), None))
src = int(torch.nonzero(deg[:1 << 16] > 0)[0].item())
dist = torch.empty(m, dtype=torch.int32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
plans = []
for k in range(K):
    plan = C.c_void_p(); _cabi.check(L.gdn_sssp_plan_create(go, p(w), 1, C.byref(plan)))
    plans.append(plan)
for rnd in range(3):
    for k, plan in enumerate(plans):
        ts = []
        for _ in range(5):
            st = _cabi.GdnStats(); _cabi.check(L.gdn_sssp_run(plan, src, 16, p(dist), C.byref(st)))
            ts.append(st.solve_ms)
        print("round %d plan %d: %s" % (rnd, k, " ".join("%.3f" % t for t in ts)), flush=True)
