#!/usr/bin/env python3
"""Every solver on NON-R-MAT graph shapes (grid / uniform random / clustered small world) against the CPU oracle, with
PageRank's roofline fraction and BFS GTEPS per shape.  usage: shapes.py [small|large] [out.json]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from gardenia_amd import _cabi, graphio, solvers
from oracle import binding as orc

L = _cabi.lib()
size = sys.argv[1] if len(sys.argv) > 1 else "small"
SHAPES = {
    "small": {"grid2d": lambda: graphio.grid2d_edges(256, 256), "uniform": lambda: graphio.uniform_edges(1 << 17, 1 << 20, 7),
              "small_world": lambda: graphio.small_world_edges(1 << 16, 16, 0.1, 7)},
    "large": {"grid2d": lambda: graphio.grid2d_edges(4096, 4096), "uniform": lambda: graphio.uniform_edges(1 << 23, 1 << 26, 7),
              "small_world": lambda: graphio.small_world_edges(1 << 22, 16, 0.1, 7)},
}[size]


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def run_shape(name, make):
    t0 = time.time()
    m, src, dst = make()
    g = graphio.build_csr_device(m, src, dst)
    del src, dst
    gi = graphio.transpose(g) if g.nnz < 5_000_000 else None
    rec = {"vertices": g.m, "edges": g.nnz, "max_out_degree": int(g.degrees().max())}
    # resident graphs for the plans
    ho, hi = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_graph_upload(g.m, g.nnz, p(g.rowptr), p(g.colidx), C.byref(ho)))
    _cabi.check(L.gdn_graph_transpose(ho, C.byref(hi)))
    if gi is None:
        mm, nn = C.c_int32(), C.c_uint64()
        _cabi.check(L.gdn_graph_info(hi, C.byref(mm), C.byref(nn), None, None))
        rp, ci = np.empty(g.m + 1, np.uint64), np.empty(nn.value, np.int32)
        _cabi.check(L.gdn_graph_download(hi, p(rp), p(ci)))
        gi = graphio.CSR(g.m, rp, ci)
    deg = g.degrees()
    s = graphio.first_nonisolated(g)
    G = solvers.Graph(csr=g, in_csr=gi)
    rec["build_s"] = time.time() - t0
    lap_t = [time.time()]

    def lap(what):
        now = time.time()
        rec.setdefault("seconds", {})[what] = round(now - lap_t[0], 2)
        lap_t[0] = now
    # ---- BFS: resident plan (dense levels where the chooser takes them) and the drop-in, exact
    want = orc.bfs_serial(g, s)
    d_dist = C.c_void_p()
    _cabi.check(L.gdn_dev_alloc(4 * g.m, C.byref(d_dist)))
    bplan = C.c_void_p()
    _cabi.check(L.gdn_bfs_plan_create(ho, hi, 1, C.byref(bplan)))
    best = None
    for _ in range(3):
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_bfs_run(bplan, s, d_dist, C.byref(st)))
        best = st.solve_ms if best is None else min(best, st.solve_ms)
    got = np.empty(g.m, np.int32)
    _cabi.check(L.gdn_dev_download(p(got), d_dist, 4 * g.m))
    assert np.array_equal(got, want), name + " BFS plan"
    L.gdn_bfs_plan_free(bplan)
    dist = np.full(g.m, solvers.MYINFINITY, np.int32)
    solvers.BFSSolver(G, s, dist)
    assert np.array_equal(dist, want), name + " BFS"
    rec["bfs"] = {"ms": best, "levels": st.iterations, "gteps": st.edges_traversed / best / 1e6}
    lap("bfs")
    # ---- SSSP: resident plan with dense sweeps + the drop-in, exact
    rng = np.random.default_rng(5)
    w = rng.integers(1, 256, g.nnz).astype(np.int32)
    want = orc.sssp_dijkstra(g, w, s)
    sp = solvers.ResidentSSSP(G, w, dense=True)
    t1 = None
    for _ in range(2):
        d2, st = sp.run(s, 16)
        t1 = st["solve_ms"] if t1 is None else min(t1, st["solve_ms"])
    sp.close()
    assert np.array_equal(d2, want), name + " SSSP plan"
    rec["sssp_u1_255_delta16"] = {"ms": t1, "phases": st["iterations"]}
    lap("sssp")
    # ---- PageRank: drop-in to convergence vs the oracle, then the resident plan's iteration time / roofline
    want, it, trace = orc.pr(gi, deg)
    scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st = solvers.PRSolver(G, scores)
    assert st["iterations"] == it, (name, st["iterations"], it)
    assert float((np.abs(scores - want) / want).max()) < 1e-4, name + " PR"
    d_deg, d_s, d_c0, d_c1, d_diff = (C.c_void_p() for _ in range(5))
    for d, n in ((d_deg, 4 * g.m), (d_s, 4 * g.m), (d_c0, 4 * g.m + 16), (d_c1, 4 * g.m + 16), (d_diff, 8)):
        _cabi.check(L.gdn_dev_alloc(n, C.byref(d)))
    _cabi.check(L.gdn_dev_upload(d_deg, p(deg.astype(np.int32)), 4 * g.m))
    plan = C.c_void_p()
    _cabi.check(L.gdn_pr_plan_create(hi, d_deg, g.m, 0, _cabi.GDN_LAYOUT_AUTO, C.byref(plan)))
    lay, lg = C.c_int32(), C.c_int32()
    _cabi.check(L.gdn_pr_plan_layout(plan, C.byref(lay), C.byref(lg)))
    nh, he = C.c_int32(0), C.c_uint64(0)
    _cabi.check(L.gdn_pr_plan_hubs(plan, C.byref(nh), C.byref(he)))
    mt, ms_, me = C.c_int32(0), C.c_int32(0), C.c_uint64(0)
    _cabi.check(L.gdn_pr_plan_mid(plan, C.byref(mt), C.byref(ms_), C.byref(me)))
    init = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    _cabi.check(L.gdn_dev_upload(d_s, p(init), 4 * g.m))
    _cabi.check(L.gdn_pr_contrib_dev(plan, d_s, d_c0, None))
    cin, cout = d_c0, d_c1
    for _ in range(3):
        _cabi.check(L.gdn_pr_pull_dev(plan, cin, d_s, cout, d_diff, 0.85, None))
        cin, cout = cout, cin
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 1, 10, None, None))
    for _ in range(10):
        _cabi.check(L.gdn_pr_pull_dev(plan, cin, d_s, cout, d_diff, 0.85, None))
        cin, cout = cout, cin
    tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 0, 0, tot, C.byref(n)))
    _cabi.check(L.gdn_pr_plan_check(plan))
    k_ms = (tot[0] + tot[1]) / max(n.value, 1)
    b = int(L.gdn_pr_iter_bytes(plan))
    rec["pagerank"] = {"iterations_to_converge": it, "layout": "pb" if lay.value == 1 else "csr", "hubs": nh.value,
                       "mid_tier_sources": ms_.value, "tier_edge_share": (he.value + me.value) / max(g.nnz, 1),
                       "ms_per_iter": k_ms, "roofline_frac": b / (k_ms * 1e-3) / 1e9 / 8000.0 if k_ms > 0 else 0.0}
    L.gdn_pr_plan_free(plan)
    for d in (d_deg, d_s, d_c0, d_c1, d_diff, d_dist):
        L.gdn_dev_free(d)
    lap("pagerank")
    # ---- SpMV: resident plan (AUTO) vs the oracle on every row
    Ax, x, y0 = rng.random(gi.nnz, dtype=np.float32), rng.random(g.m, dtype=np.float32), rng.random(g.m, dtype=np.float32)
    want = orc.spmv(gi, Ax, x, y0)
    rs = solvers.ResidentSpMV(G, Ax)
    got = rs.multiply(x, y0)
    rs.close()
    assert orc.spmv_max_rel_error(got, want) <= 5 * np.sqrt(np.finfo(np.float32).eps), name + " SpMV"
    lap("spmv")
    # ---- CC: with and without the reverse graph, labels == the oracle's (minimum vertex id per component)
    want, _ = orc.cc_sv(g)
    for GG in (G, solvers.Graph(csr=g)):
        comp = np.arange(g.m, dtype=np.int32)
        st = solvers.CCSolver(GG, comp)
        assert np.array_equal(comp, want), name + " CC"
    rec["cc"] = {"ms_without_reverse": st["solve_ms"], "components": int(len(np.unique(want)))}
    lap("cc")
    # ---- TC on the symmetrized graph == the oracle's count
    gs = graphio.build_csr_device(g.m, *graphio.csr_to_coo(g), symmetrize_=True)
    want = orc.tc(orc.tc_orient(gs))
    total, st = solvers.TCSolver(solvers.Graph(csr=gs, in_csr=gs))
    assert total == want, (name, total, want)
    total2, st2 = solvers.TCSolver(solvers.Graph(csr=gs, in_csr=gs))  # (a one-shot call: the first pays the cold start)
    assert total2 == want
    if st2["solve_ms"] < st["solve_ms"]:
        st = st2
    rec["tc"] = {"triangles": want, "count_ms": st["solve_ms"], "form": {0: "u-centric", 1: "v-centric", 2: "binary search", 3: "forward"}.get(int(st["reserved"]) & 0xFF, str(st["reserved"]))}
    lap("tc")
    # ---- BC from one source within the reference verifier's tolerance
    sc = np.zeros(g.m, np.float32)
    st = solvers.BCSolver(solvers.Graph(csr=g), s, sc)
    if rec["bfs"]["levels"] <= 600:  # the serial verifier takes minutes on thousands of levels (the large grid)
        assert orc.bc_verify(g, s, sc), name + " BC"
    rec["bc"] = {"ms": st["solve_ms"], "levels": st["iterations"]}
    lap("bc")
    L.gdn_graph_free(ho)
    L.gdn_graph_free(hi)
    rec["total_s"] = time.time() - t0
    return rec


def main():
    out = {}
    for name, make in SHAPES.items():
        out[name] = run_shape(name, make)
        print(name, json.dumps(out[name]), flush=True)
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
