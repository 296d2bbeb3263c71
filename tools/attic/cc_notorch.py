"""CC timing through the C-ABI: R-MAT scale S (directed): Afforest with the reverse graph, without it, and SV rounds."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gardenia_amd import _cabi, graphio
L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
go, gi = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None))
d = C.c_void_p(); _cabi.check(L.gdn_dev_alloc(4 * m.value, C.byref(d)))
ref = None
for name, rev, env in (("afforest + reverse graph", gi, None), ("out-edges only", None, None), ("SV rounds", None, "1")):
    if env: os.environ["GDN_CC_SV"] = env
    best = 1e9
    for _ in range(3):
        st = _cabi.GdnStats(); _cabi.check(L.gdn_cc_dev(go, rev, d, C.byref(st))); best = min(best, st.solve_ms)
    comp = np.empty(m.value, np.int32); _cabi.check(L.gdn_dev_download(comp.ctypes.data_as(C.c_void_p), d, 4 * m.value))
    if ref is None: ref = comp
    print("RMAT-%d CC %-26s %.3f ms (%d rounds) labels equal: %s components %d" % (scale, name, best, st.iterations, np.array_equal(comp, ref), len(np.unique(comp))))
