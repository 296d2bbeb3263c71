import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from gardenia_amd import _cabi, graphio
L = _cabi.lib(); dev = torch.device("cuda", 0)
go = C.c_void_p(); _cabi.check(L.gdn_rmat_build(24, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
m, nnz = C.c_int32(), C.c_uint64(); _cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None)); m, nnz = m.value, nnz.value
torch.manual_seed(5)
w = torch.randint(1, 256, (nnz,), dtype=torch.int32, device=dev)
deg = torch.empty(m, dtype=torch.int32, device=dev); _cabi.check(L.gdn_graph_degrees_dev(go, C.c_void_p(deg.data_ptr()), None))
src = int(torch.nonzero(deg[:1 << 16] > 0)[0].item())
dist = torch.empty(m, dtype=torch.int32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
plan = C.c_void_p(); _cabi.check(L.gdn_sssp_plan_create(go, p(w), 1, C.byref(plan)))
for delta in (16, 64):
    for name, fn in (("worklist", lambda st: L.gdn_sssp_dev(go, p(w), src, delta, p(dist), C.byref(st))), ("plan", lambda st: L.gdn_sssp_run(plan, src, delta, p(dist), C.byref(st)))):
        best = None
        for _ in range(3):
            st = _cabi.GdnStats(); _cabi.check(fn(st))
            best = st.solve_ms if best is None else min(best, st.solve_ms)
        print("delta %d %s: %.3f ms (%d phases) checksum %d" % (delta, name, best, st.iterations, int(dist.clamp(max=10**9).to(torch.int64).sum().item())))
