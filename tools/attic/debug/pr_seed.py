"""PageRank of one fuzz seed through the one-shot drop-in under several builder settings, against the oracle."""
import os, sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests", "aids"))
import numpy as np
from gardenia_amd import graphio, solvers
from oracle import binding as orc
import fuzz_parity as fz

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
g = fz.random_graph(rng)
gi = graphio.transpose(g)
G = solvers.Graph(csr=g, in_csr=gi, need_reverse=True)
deg = np.diff(g.rowptr.astype(np.int64))
want, it, _ = orc.pr(gi, deg.astype(np.int32))
print("seed", seed, "m", g.m, "nnz", g.nnz, "oracle iterations", it, "max out-degree", deg.max(), "sources deg>=4:", int((deg >= 4).sum()))
base = dict(GDN_PR_LAYOUT="p", GDN_PB_HUB_MIN_NNZ="1", GDN_PB_TRACE="1")
order = sys.argv[2].split(",") if len(sys.argv) > 2 else ["new", "old", "notiers", "nosquish"]
variants = {"new": {}, "old": {"GDN_PB_BUILDER": "old"}, "notiers": {"GDN_PB_HUBS": "0"}, "nosquish": {"GDN_PR_SQUISH": "0"}}
for name in order:
  extra = variants[name]
  try:
    for k in ("GDN_PB_BUILDER", "GDN_PB_HUBS", "GDN_PR_SQUISH"):
        os.environ.pop(k, None)
    os.environ.update(base)
    os.environ.update(extra)
    s = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st = solvers.PRSolver(G, s)
    rel = np.abs(s - want) / np.maximum(np.abs(want), 1e-30)
    bad = np.nonzero(rel > 1e-4)[0]
    print("%-16s iterations %d (oracle %d) max rel %.3e, %d vertices off: %s" % (name, st["iterations"], it, rel.max(), len(bad), bad[:12]), flush=True)
    if len(bad):
        indeg = np.diff(gi.rowptr.astype(np.int64))
        print("    in-degrees of the first:", indeg[bad[:12]], "got", s[bad[:6]], "want", want[bad[:6]])
  except Exception as e:
    print("%-16s FAILED: %s" % (name, str(e)[:200]), flush=True)
