"""BFS on an nx x nx lattice through the C-ABI (high diameter: the per-level cost is what is measured)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gardenia_amd import _cabi, graphio, solvers
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m, src, dst = graphio.grid2d_edges(nx, nx)
g = graphio.build_csr_device(m, src, dst)
G = solvers.Graph(csr=g, in_csr=g)
bfs = solvers.ResidentBFS(G, dense=True)
for s in (0, m // 2 + nx // 2):
    best = None
    for _ in range(3):
        dist, st = bfs.run(s)
        best = st["solve_ms"] if best is None else min(best, st["solve_ms"])
    print("grid %dx%d BFS from %d: %.3f ms, %d levels = %.2f us/level, checksum %d" % (nx, nx, s, best, st["iterations"], 1e3 * best / st["iterations"], int(dist.astype(np.int64).sum())))
bfs.close()
