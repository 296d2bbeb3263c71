import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from gardenia_amd import graphio, solvers
scales = [int(x) for x in sys.argv[1:]] or [18, 20, 22]
for scale in scales:
    g = graphio.rmat_graph(scale, 16, seed=3)
    G = solvers.Graph(csr=g, need_reverse=True)
    for lay in ("csr", "pb"):
        os.environ["GDN_PR_LAYOUT"] = lay
        for rep in range(2):
            sc = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
            st = solvers.PRSolver(G, sc)
        print("scale %d %s: iterations %d solve %.2f ms prep %.2f ms h2d %.2f ms  (nnz %d)" % (scale, lay, st["iterations"], st["solve_ms"], st["prep_ms"], st["h2d_ms"], g.nnz))
