"""BC on a high-diameter graph (a chain of n vertices / an n-vertex-per-side lattice): ms with the fused backward levels
off and on.  usage: python tools/bc_chain.py [n_chain] [lattice_side]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gardenia_amd import graphio, solvers  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
side = int(sys.argv[2]) if len(sys.argv) > 2 else 512
chain = graphio.build_csr(n, np.arange(n - 1, dtype=np.int64), np.arange(1, n, dtype=np.int64))
nv, s, d = graphio.grid2d_edges(side, side)
lattice = graphio.build_csr(nv, s, d)
for name, g in (("chain %d" % n, chain), ("lattice %dx%d" % (side, side), lattice)):
    G = solvers.Graph(csr=g)
    ref = None
    for mode in ("0", None):
        if mode is None:
            os.environ.pop("GDN_BC_BACK_NF", None)
        else:
            os.environ["GDN_BC_BACK_NF"] = mode
        best = 1e30
        for _ in range(3):
            sc = np.zeros(g.m, np.float32)
            st = solvers.BCSolver(G, 0, sc)
            best = min(best, st["solve_ms"])
        if ref is None:
            ref = sc
        print("%-20s back_nf %-8s %9.2f ms  levels %d  same bits %s" % (name, mode or "default", best, st["iterations"],
                                                                      bool(np.array_equal(ref.view(np.uint32), sc.view(np.uint32)))))
