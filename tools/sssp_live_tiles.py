#!/usr/bin/env python3
"""would tile-granular skipping help SSSP's dense sweeps?  CPU measurement (numpy, no GPU):

   python3 tools/sssp_live_tiles.py [scale 22] [log_chunk 15] [log_bin 14]

Bellman-Ford sweeps (Jacobi: every sweep relaxes all edges with the distances of the sweep before) on R-MAT(scale) x 16 with
U[1,255] weights from the first non-isolated vertex -- the dense phase of gdn_sssp_run without its list phases.  An edge (s, t) can
improve t in sweep k + 1 only if dist[s] changed in sweep k; a TILE (source chunk of 2^log_chunk ids x destination bin of
2^log_bin rows, the unit the blocked sweep layout streams) must be read in sweep k + 1 iff one of the sources that HAVE an edge in
it improved in sweep k.  Per sweep: improved vertices, the share of chunks that hold one, the share of the EDGES that lie in
live tiles (what a tile-skipping sweep would still stream), and the share of edges whose own source improved (the floor of any
source-granular scheme).  Chunk / bin sizes scale with the graph like the plan's (2^15 / 2^14 at RMAT-24: 512 x 1024 tiles)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import graphio

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
lc = int(sys.argv[2]) if len(sys.argv) > 2 else 15 - (24 - scale)
lb = int(sys.argv[3]) if len(sys.argv) > 3 else 14 - (24 - scale)
t0 = time.time()
g = graphio.rmat_graph(scale, 16)
m, nnz = g.m, g.nnz
src = np.repeat(np.arange(m, dtype=np.int64), np.diff(g.rowptr.astype(np.int64)))
dst = g.colidx.astype(np.int64)
w = np.random.default_rng(5).integers(1, 256, nnz, dtype=np.int64)
order = np.argsort(dst, kind="stable")  # in-CSR order: reduceat over destination rows
src_i, w_i, dst_i = src[order], w[order], dst[order]
starts = np.flatnonzero(np.r_[True, dst_i[1:] != dst_i[:-1]])
rows = dst_i[starts]
tile = (src >> lc) * ((m + (1 << lb) - 1) >> lb) + (dst >> lb)
ntiles = int(tile.max()) + 1
print("# R-MAT %d x 16: %d vertices, %d edges, chunks of 2^%d sources x bins of 2^%d rows = %d x %d tiles (%.1f s to build)" % (
    scale, m, nnz, lc, lb, (m + (1 << lc) - 1) >> lc, (m + (1 << lb) - 1) >> lb, time.time() - t0))
INF = np.int64(1) << 40
dist = np.full(m, INF)
s0 = int(graphio.first_nonisolated(g))
dist[s0] = 0
improved = np.zeros(m, bool)
improved[s0] = True
print("| sweep | vertices improved by the sweep before | chunks holding one | edges in live tiles | edges whose own source improved |")
print("|---|---|---|---|---|")
k = 0
while improved.any():
    k += 1
    live_src_edge = improved[src]
    live_tiles = np.zeros(ntiles, bool)
    live_tiles[tile[live_src_edge]] = True
    share_tiles = float(live_tiles[tile].mean())
    chunks = np.unique(np.flatnonzero(improved) >> lc).size / float(((m + (1 << lc) - 1) >> lc))
    print("| %d | %d (%.1f %% of |V|) | %.1f %% | %.1f %% | %.1f %% |" % (k, int(improved.sum()), 100.0 * improved.mean(), 100.0 * chunks,
                                                                 100.0 * share_tiles, 100.0 * float(live_src_edge.mean())), flush=True)
    cand = dist[src_i] + w_i
    best = np.minimum.reduceat(cand, starts)
    new = dist.copy()
    new[rows] = np.minimum(dist[rows], best)
    improved = new < dist
    dist = new
print("# %d sweeps to the fixpoint, %d vertices reached" % (k, int((dist < INF).sum())))
