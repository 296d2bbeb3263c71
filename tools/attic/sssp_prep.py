"""gdn_sssp_dev one-shot on R-MAT <scale>: prep (the blocked layout built inside the call) and solve, unit and U[1,255] weights."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from gardenia_amd import _cabi, graphio
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L = _cabi.lib(); dev = torch.device("cuda", 0)
go = C.c_void_p(); _cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), None))
m, nnz = C.c_int32(), C.c_uint64(); _cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None)); m, nnz = m.value, nnz.value
deg = torch.empty(m, dtype=torch.int32, device=dev); _cabi.check(L.gdn_graph_degrees_dev(go, C.c_void_p(deg.data_ptr()), None))
src = int(torch.nonzero(deg[:1 << 16] > 0)[0].item())
dist = torch.empty(m, dtype=torch.int32, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(5)
p = lambda t: C.c_void_p(t.data_ptr())
for name, w, delta in (("unit", torch.ones(nnz, dtype=torch.int32, device=dev), 1), ("u1_255", torch.randint(1, 256, (nnz,), dtype=torch.int32, device=dev, generator=gen), 16)):
    for rep in range(3):
        st = _cabi.GdnStats(); _cabi.check(L.gdn_sssp_dev(go, p(w), src, delta, p(dist), C.byref(st)))
        print("scale %d %s: prep %.2f ms solve %.2f ms phases %d checksum %d" % (scale, name, st.prep_ms, st.solve_ms, st.iterations, int(dist.clamp(max=10**9).to(torch.int64).sum().item())), flush=True)
