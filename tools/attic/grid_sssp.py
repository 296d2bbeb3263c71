"""SSSP on an nx x nx lattice with U[1,255] weights through the C-ABI (thousands of light buckets)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gardenia_amd import _cabi, graphio, solvers
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
delta = int(sys.argv[2]) if len(sys.argv) > 2 else 16
m, src, dst = graphio.grid2d_edges(nx, nx)
g = graphio.build_csr_device(m, src, dst)
w = np.random.default_rng(5).integers(1, 256, g.nnz).astype(np.int32)
sp = solvers.ResidentSSSP(solvers.Graph(csr=g), w, dense=True)
for s in (0,):
    best = None
    for _ in range(2):
        dist, st = sp.run(s, delta)
        best = st["solve_ms"] if best is None else min(best, st["solve_ms"])
    print("grid %dx%d SSSP delta %d from %d: %.1f ms, %d passes = %.2f us/pass, checksum %d" % (nx, nx, delta, s, best, st["iterations"], 1e3 * best / st["iterations"], int(dist.astype(np.int64).sum())))
sp.close()
