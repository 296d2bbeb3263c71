#!/usr/bin/env python3
"""SSSP solve time against the bucket width: python tools/sssp_delta_sweep.py rmat <scale> | grid <nx> | uniform <log2 m> <deg>
U[1,255] weights, one resident plan, every delta solved 3 times (median), distances compared with the first delta's."""
import ctypes as C
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import _cabi, graphio, solvers

kind = sys.argv[1]
if kind == "grid":
    nx = int(sys.argv[2])
    m, src, dst = graphio.grid2d_edges(nx, nx)
    g = graphio.build_csr_device(m, src, dst)
    s = 0
elif kind == "uniform":
    lm, deg = int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(3)
    m = 1 << lm
    src = rng.integers(0, m, m * deg, dtype=np.int64).astype(np.int32)
    dst = rng.integers(0, m, m * deg, dtype=np.int64).astype(np.int32)
    g = graphio.build_csr_device(m, src, dst)
    s = 0
else:
    src, dst = graphio.rmat_edges(int(sys.argv[2]), 16)
    g = graphio.build_csr_device(1 << int(sys.argv[2]), src.astype(np.int32), dst.astype(np.int32))
    s = graphio.first_nonisolated(g)
w = np.random.default_rng(5).integers(1, 256, g.nnz).astype(np.int32)
sp = solvers.ResidentSSSP(solvers.Graph(csr=g), w, dense=True)
ref = None
for delta in (16, 32, 64, 128, 256, 512, 1024, 4096):
    ts = []
    for _ in range(3):
        dist, st = sp.run(s, delta)
        ts.append(st["solve_ms"])
    c = zlib.crc32(dist.tobytes())
    if ref is None:
        ref = c
    print("%s delta %5d: %9.3f ms  %6d phases  relaxed/traversed %.2f  %s" % (
        " ".join(sys.argv[1:]), delta, float(np.median(ts)), st["iterations"], st["last_error"] / max(1, st["edges_traversed"]),
        "same distances" if c == ref else "DISTANCES DIFFER"), flush=True)
sp.close()
