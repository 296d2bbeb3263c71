#!/usr/bin/env python3
"""How evenly do the in-edges of R-MAT scale S fall into 256 equal id ranges (the bins of the binned top-down BFS level)?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gardenia_amd import _cabi, graphio
L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
go, gi = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(gi, C.byref(m), C.byref(nnz), None, None)); m = m.value
d = C.c_void_p(); _cabi.check(L.gdn_dev_alloc(4 * m, C.byref(d)))
_cabi.check(L.gdn_graph_degrees_dev(gi, d, None))
deg = np.empty(m, np.int32); _cabi.check(L.gdn_dev_download(deg.ctypes.data_as(C.c_void_p), d, 4 * m))
nb = 256
s = deg.astype(np.int64).reshape(nb, -1).sum(1)
print("in-edges per bin: max %.2f %% of all, mean %.2f %%, top five %s" % (100 * s.max() / s.sum(), 100 / nb, np.sort(s)[::-1][:5] * 100.0 / s.sum()))
