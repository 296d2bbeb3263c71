"""One traced SSSP solve on RMAT-<scale> (GDN_SSSP_TRACE=1 in the environment prints the phases).
usage: sssp_trace.py [scale] [delta] [unit|rand] [plan|dev]"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from gardenia_amd import _cabi, graphio
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
delta = int(sys.argv[2]) if len(sys.argv) > 2 else 16
kind = sys.argv[3] if len(sys.argv) > 3 else "rand"
mode = sys.argv[4] if len(sys.argv) > 4 else "plan"
L = _cabi.lib(); dev = torch.device("cuda", 0)
go = C.c_void_p(); _cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
m, nnz = C.c_int32(), C.c_uint64(); _cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None)); m, nnz = m.value, nnz.value
torch.manual_seed(5)
w = torch.randint(1, 256, (nnz,), dtype=torch.int32, device=dev) if kind == "rand" else torch.ones(nnz, dtype=torch.int32, device=dev)
deg = torch.empty(m, dtype=torch.int32, device=dev); _cabi.check(L.gdn_graph_degrees_dev(go, C.c_void_p(deg.data_ptr()), None))
src = int(torch.nonzero(deg[:1 << 16] > 0)[0].item())
dist = torch.empty(m, dtype=torch.int32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
plan = C.c_void_p(); _cabi.check(L.gdn_sssp_plan_create(go, p(w), 1 if mode == "plan" else 0, C.byref(plan)))
reps = int(os.environ.get("REPS", "1"))
for _ in range(reps):
    st = _cabi.GdnStats(); _cabi.check(L.gdn_sssp_run(plan, src, delta, p(dist), C.byref(st)))
    print("RMAT-%d %s delta %d %s: %.3f ms (%d phases) edges %d checksum %d" % (scale, kind, delta, mode, st.solve_ms, st.iterations, st.edges_traversed, int(dist.clamp(max=10**9).to(torch.int64).sum().item())), flush=True)
