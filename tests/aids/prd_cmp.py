#!/usr/bin/env python3
"""Delta PageRank against the oracle on an R-MAT graph of scale S (debug aid: crc of the scores, per-iteration trace)."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gardenia_amd import graphio, solvers
from oracle import binding as orc
scale = int(sys.argv[1])
push_div = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = graphio.rmat_graph(scale, 16, seed=3)
gi = graphio.transpose(g)
want, it, wtr = orc.pr_delta(gi, g, push_div=push_div)
r = solvers.ResidentPRDelta(solvers.Graph(csr=g, in_csr=gi))
for rep in range(2):
    s = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st, tr = r.run(s, push_div=push_div)
    rel = np.abs(s - want) / np.maximum(np.abs(want), 1e-30)
    print(os.environ.get("GARDENIA_HIP_LIB", "default")[-30:], scale, st["iterations"], it, "crc %08x" % zlib.crc32(s.tobytes()),
          "max rel %.3e" % rel.max(), "n>1e-5: %d" % (rel > 1e-5).sum(), "items eq", np.array_equal(tr["items"], wtr["items"]),
          tr["mode"].tolist(), tr["masked"].tolist())
print(" diff trace:", " ".join(x.hex() for x in tr["diff"]))
print(" items:", tr["items"].tolist(), "oracle", wtr["items"].tolist())
if len(sys.argv) > 3:
    np.save(sys.argv[3], s)
