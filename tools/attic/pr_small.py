"""PageRank on small graphs: gdn_pr's solve time with the whole solve in one cooperative launch (pr_fused_kernel) against
the per-iteration loop (merge-path CSR plan).  usage: python tools/pr_small.py [scale ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gardenia_amd import graphio, solvers  # noqa: E402

# GDN_PR_BATCH=1 in the environment: the loop reads the L1 change back after every iteration (round 1's form)
for scale in [int(a) for a in sys.argv[1:]] or [10, 14, 16, 18]:
    g = graphio.rmat_graph(scale, 16, seed=5)
    G = solvers.Graph(csr=g, need_reverse=True)
    row = []
    for mode in ("0", "1"):
        os.environ["GDN_PR_FUSED"] = mode
        best = 1e30
        for _ in range(3):
            s = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
            st = solvers.PRSolver(G, s)
            best = min(best, st["solve_ms"])
        row.append((best, st["iterations"]))
    print("RMAT-%d (%d edges): loop %.3f ms / %d iterations = %.1f us each; fused %.3f ms / %d = %.1f us each" % (
        scale, g.nnz, row[0][0], row[0][1], 1e3 * row[0][0] / row[0][1], row[1][0], row[1][1], 1e3 * row[1][0] / row[1][1]))
