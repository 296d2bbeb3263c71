#!/usr/bin/env python3
"""Medium-size parity sweep: seeded R-MAT graphs of scales 18..22 through the resident plans (BFS, SSSP with several weight ranges
and deltas, CC, PageRank, SpMV, TC) against the CPU oracle.  python tools/mid_sweep.py [scales] [seeds]
(test infrastructure: the oracle is the checker)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gardenia_amd import graphio, solvers
from oracle import binding as orc

scales = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["18", "20", "21"])]
seeds = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["101", "202", "303"])]
bad = 0
for scale in scales:
    for seed in seeds:
        t0 = time.time()
        rng = np.random.default_rng(seed)
        src, dst = graphio.rmat_edges(scale, 16, seed=seed)
        g = graphio.build_csr_device(1 << scale, src.astype(np.int32), dst.astype(np.int32))
        gi = graphio.transpose(g)
        G = solvers.Graph(csr=g, in_csr=gi)
        deg = g.degrees()
        live = np.nonzero(deg > 0)[0]
        sources = [int(live[0]), int(np.argmax(deg)), int(rng.choice(live))]
        msgs = []
        rb = solvers.ResidentBFS(G, dense=True)
        for s in sources:
            d, _ = rb.run(s)
            if not np.array_equal(d, orc.bfs_serial(g, s)):
                msgs.append(f"BFS source {s}")
        rb.close()
        for wlo, whi, deltas in ((1, 255, (16, 1)), (1, 15, (3,)), (1, 1, (1,)), (0, 3, (2,)), (1, 60000, (4096,))):
            w = rng.integers(wlo, whi + 1, g.nnz).astype(np.int32)
            rs = solvers.ResidentSSSP(G, w, dense=True)
            for s in sources[:2]:
                want = orc.sssp_dijkstra(g, w, s)
                for delta in deltas:
                    d, _ = rs.run(s, delta)
                    if not np.array_equal(d, want):
                        msgs.append(f"SSSP weights [{wlo},{whi}] source {s} delta {delta}: {int((d != want).sum())} differ")
            rs.close()
        want, _ = orc.cc_sv(graphio.symmetrize(g))
        for GG in (G, solvers.Graph(csr=g)):
            comp = np.arange(g.m, dtype=np.int32)
            solvers.CCSolver(GG, comp)
            if not np.array_equal(comp, want):
                msgs.append("CC")
        want, it, otr = orc.pr(gi, deg.astype(np.int32))
        s_ = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        st = solvers.PRSolver(G, s_)
        if st["iterations"] != it or not np.allclose(s_, want, rtol=1e-4, atol=0):
            rel = np.abs(s_ - want) / np.maximum(np.abs(want), 1e-30)
            msgs.append(f"PR iterations {st['iterations']} vs {it}, {int((rel > 1e-4).sum())} rows beyond 1e-4 (max {rel.max():.3g}, min in-degree of those {int(np.diff(gi.rowptr.astype(np.int64))[rel > 1e-4].min()) if (rel > 1e-4).any() else 0})")
        Ax = rng.random(g.nnz, dtype=np.float32)
        x = rng.random(g.m, dtype=np.float32)
        y0 = np.zeros(g.m, np.float32)
        sp = solvers.ResidentSpMV(G, Ax, layout=1)
        y = sp.multiply(x, y0)
        sp.close()
        if orc.spmv_max_rel_error(y, orc.spmv(gi, Ax, x, y0)) > 5 * np.sqrt(np.finfo(np.float32).eps):
            msgs.append("SpMV")
        gs = graphio.symmetrize(g)
        got, _ = solvers.TCSolver(solvers.Graph(csr=gs, symmetrize=True))
        if got != orc.tc(orc.tc_orient(gs)):
            msgs.append("TC")
        bad += len(msgs)
        print(f"scale {scale} seed {seed}: |E| {g.nnz}: {'ok' if not msgs else 'MISMATCH ' + '; '.join(msgs)}  ({time.time() - t0:.0f} s)", flush=True)
print("mid sweep:", "all equal to the oracle" if bad == 0 else f"{bad} mismatches")
sys.exit(1 if bad else 0)
