"""Re-run the SSSP part of one fuzz seed (tests/aids/fuzz_parity.py) through the resident plan and the drop-in."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "aids"))
import numpy as np
import fuzz_parity as fp
from gardenia_amd import graphio, solvers
from oracle import binding as orc
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
g = fp.random_graph(rng)
m = g.m
source = int(rng.integers(0, m))
wmax = int(rng.choice([2, 16, 256]))
w = rng.integers(1, wmax, g.nnz).astype(np.int32)
delta = int(rng.choice([1, 3] if wmax <= 16 else [16, 64]))
print("seed", seed, "m", m, "nnz", g.nnz, "source", source, "wmax", wmax, "delta", delta, flush=True)
want = orc.sssp_dijkstra(g, w, source)
G = solvers.Graph(csr=g)
rs = solvers.ResidentSSSP(G, w, dense=True)
d, st = rs.run(source, delta)
bad = np.nonzero(d != want)[0]
print("plan: mismatches", len(bad), bad[:10], d[bad[:10]], want[bad[:10]], "phases", st["iterations"], flush=True)
rs.close()
d2 = np.full(m, solvers.K_DIST_INF, np.int32)
solvers.SSSPSolver(G, source, w, d2, delta)
print("drop-in: mismatches", int((d2 != want).sum()))
