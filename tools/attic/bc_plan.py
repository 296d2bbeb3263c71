"""BC from one source through the resident plan on R-MAT: python tools/bc_plan.py [scale] [runs]  (for rocprofv3 runs)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gardenia_amd import _cabi, graphio  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L = _cabi.lib()
dev = torch.device("cuda", 0)
go, gi = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
m = C.c_int32()
_cabi.check(L.gdn_graph_info(go, C.byref(m), None, None, None))
m = m.value
deg = torch.empty(m, dtype=torch.int32, device=dev)
_cabi.check(L.gdn_graph_degrees_dev(go, C.c_void_p(deg.data_ptr()), None))
src = int(torch.nonzero(deg[:1 << 16] > 0)[0].item())
sc = torch.zeros(m, dtype=torch.float32, device=dev)
bplan = C.c_void_p()
_cabi.check(L.gdn_bc_plan_create(go, gi, C.byref(bplan)))
best = 1e30
for _ in range(runs):
    sc.zero_()
    st = _cabi.GdnStats()
    _cabi.check(L.gdn_bc_run(bplan, src, C.c_void_p(sc.data_ptr()), C.byref(st)))
    best = min(best, st.solve_ms)
print("RMAT-%d BC plan from %d: %.3f ms best of %d, %d levels, checksum %.9g" % (scale, src, best, runs, st.iterations, float(sc.double().sum().item())))
L.gdn_bc_plan_free(bplan)
