#!/usr/bin/env python3
"""Debug aid: delta PageRank on the fuzz graph of a seed, per push form, against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import fuzz_parity as f
from gardenia_amd import graphio, solvers
from oracle import binding as orc
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
g = f.random_graph(rng)
gi = graphio.transpose(g)
want, it, wtr = orc.pr_delta(gi, g, push_div=8)
deg, indeg = np.diff(g.rowptr.astype(np.int64)), np.diff(gi.rowptr.astype(np.int64))
print("m", g.m, "nnz", g.nnz, "oracle iters", it, "mode", wtr["mode"].tolist(), "items", wtr["items"].tolist())
for div in ("1", "64", "1000000000"):
    os.environ["GDN_PRD_PUSH_DIV"] = div
    r = solvers.ResidentPRDelta(solvers.Graph(csr=g, in_csr=gi))
    s = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st, tr = r.run(s)
    r.close()
    rel = np.abs(s - want) / np.maximum(np.abs(want), 1e-30)
    bad = np.nonzero(rel > 1e-4)[0]
    print("push_div", div, "iters", st["iterations"], "masked", tr["masked"].tolist(), "items", tr["items"].tolist())
    print("   n>1e-4:", len(bad), "max rel %.3e" % rel.max(), "bad ids", bad[:10].tolist(), "indeg", indeg[bad[:10]].tolist(),
          "outdeg", deg[bad[:10]].tolist(), "got", s[bad[:5]].tolist(), "want", want[bad[:5]].tolist())
    print("   diff trace dev:", np.abs(tr["diff"] - wtr["diff"]).max())
