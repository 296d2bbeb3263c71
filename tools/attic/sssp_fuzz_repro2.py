"""Repro of a fuzz_parity SSSP-plan mismatch: python tools/sssp_fuzz_repro2.py <seed>  (same draws as tests/aids/fuzz_parity.py check())"""
import importlib.util, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
spec = importlib.util.spec_from_file_location("fz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "aids", "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from gardenia_amd import graphio, solvers
from oracle import binding as orc
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
g = fz.random_graph(rng)
m = g.m
source = int(rng.integers(0, m))
wmax = int(rng.choice([2, 16, 256]))
w = rng.integers(1, wmax, g.nnz).astype(np.int32)
delta = int(rng.choice([1, 3] if wmax <= 16 else [16, 64]))
print("m", m, "nnz", g.nnz, "source", source, "wmax", wmax, "delta", delta, "max out-degree", int(np.diff(g.rowptr.astype(np.int64)).max()))
want = orc.sssp_dijkstra(g, w, source)
G = solvers.Graph(csr=g)
rs = solvers.ResidentSSSP(G, w, dense=True)
d, st = rs.run(source, delta)
bad = np.nonzero(d != want)[0]
print("mismatches", len(bad), "phases", st["iterations"])
for v in bad[:10]:
    print("  vertex", int(v), "got", int(d[v]), "want", int(want[v]))
rs.close()
