import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
from gardenia_amd import _cabi, graphio
L = _cabi.lib()
go, gi = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(27, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(gi, C.byref(m), C.byref(nnz), None, None))
m = m.value
dev = torch.device("cuda", 0)
od = torch.empty(m, dtype=torch.int32, device=dev); idg = torch.empty(m, dtype=torch.int32, device=dev)
_cabi.check(L.gdn_graph_degrees_dev(go, C.c_void_p(od.data_ptr()), None))
_cabi.check(L.gdn_graph_degrees_dev(gi, C.c_void_p(idg.data_ptr()), None))
a, b = od > 0, idg > 0
print("m", m, "out>0", int(a.sum()), "in>0", int(b.sum()), "union", int((a | b).sum()), "both", int((a & b).sum()))
