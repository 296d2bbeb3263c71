#!/usr/bin/env python3
"""Per-kernel measurements for the remaining BASELINE.json configs (device-built R-MAT stand-ins:
soc-LiveJournal1 / com-Orkut are not in the repo and there is no network).

  config 2  PageRank pull      RMAT-22 x16 (LJ-sized: 4.2 M vertices, 65 M edges), solve to eps=1e-4
  config 3  SpMV fp32          RMAT-25 x16, Ax/x ~ U(0,1) and the constants of src/spmv/main.cc
  config 4  Triangle counting  RMAT-21 x16 symmetrized (Orkut stand-in), device orientation
  +         BFS, SSSP (unit and U[1,255] weights), CC on RMAT-24 x16

Every number is measured with the graph resident (the reference's Timer boundary); algorithmic bytes
follow SURVEY 8(d).  Prints one JSON object; tools/ is measurement scaffolding, not product code.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gardenia_amd import _cabi, graphio

L = _cabi.lib()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
HBM = 8000.0


def info(h):
    m, nnz = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(h, C.byref(m), C.byref(nnz), None, None))
    return m.value, nnz.value


def build(scale, ef=16, want_out=True, want_in=True):
    go, gi = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build(scale, ef, graphio.K_RAND_SEED, 1, C.byref(go) if want_out else None,
                                 C.byref(gi) if want_in else None))
    return go, gi


def ptr(t):
    return C.c_void_p(t.data_ptr())


def first_sources(out_deg, n=3):
    return torch.nonzero(out_deg[:1 << 16] > 0)[:n].flatten().tolist()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pr-scale", type=int, default=22)
    ap.add_argument("--spmv-scale", type=int, default=25)
    ap.add_argument("--tc-scale", type=int, default=21)
    ap.add_argument("--trav-scale", type=int, default=24)
    args = ap.parse_args()
    res = {}

    # ---------------- PageRank to convergence (config 2 stand-in)
    go, gi = build(args.pr_scale)
    m, nnz = info(gi)
    deg = torch.empty(m, dtype=torch.int32, device=dev)
    _cabi.check(L.gdn_graph_degrees_dev(go, ptr(deg), None))
    for layout, name in ((0, "csr"), (1, "pb")):
        plan = C.c_void_p()
        _cabi.check(L.gdn_pr_plan_create(gi, ptr(deg), m, 0, layout, C.byref(plan)))
        scores = torch.full((m,), 1.0 / m, dtype=torch.float32, device=dev)
        c = [torch.zeros(m, dtype=torch.float32, device=dev) for _ in range(2)]
        diff = torch.zeros(1, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _cabi.check(L.gdn_pr_contrib_dev(plan, ptr(scores), ptr(c[0]), None))
        it = 0
        for it in range(100):
            _cabi.check(L.gdn_pr_pull_dev(plan, ptr(c[it & 1]), ptr(scores), ptr(c[(it + 1) & 1]), ptr(diff), 0.85, None))
            if float(diff.item()) < 1e-4:
                break
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        b = int(L.gdn_pr_iter_bytes(plan))
        res[f"pr_rmat{args.pr_scale}_{name}"] = {
            "vertices": m, "edges": nnz, "iterations": it + 1, "solve_ms": dt * 1e3, "ms_per_iter": dt * 1e3 / (it + 1),
            "gteps": nnz * (it + 1) / dt / 1e9, "algorithmic_GBps": b * (it + 1) / dt / 1e9,
            "roofline_frac": b * (it + 1) / dt / 1e9 / HBM}
        L.gdn_pr_plan_free(plan)
    # delta PageRank (src/pr/delta.cu) to ITS stop on the same graph.  Byte model per sweep over all
    # edges (pull, or a push run as a masked pull): 8(m+1) + 8 nnz + 8 m (the pattern SpMV) + 21 m (contrib r/w + degree,
    # update: sums r/w, scores r/w, delta w, flag w, degree r)
    dplan = C.c_void_p()
    _cabi.check(L.gdn_pr_delta_plan_create(gi, go, _cabi.GDN_LAYOUT_AUTO, C.byref(dplan)))
    for _ in range(2):
        scores = torch.full((m,), 1.0 / m, dtype=torch.float32, device=dev)
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_pr_delta_run(dplan, ptr(scores), 0.85, 1e-4, 1e-3, 100, 8, C.byref(st)))
    n = C.c_int32()
    mode = (C.c_int32 * 100)()
    _cabi.check(L.gdn_pr_delta_trace(dplan, 100, C.byref(n), None, None, mode))
    sweeps = sum(1 for i in range(n.value) if mode[i] != 1)
    b = (8 * (m + 1) + 8 * nnz + 8 * m + 21 * m) * sweeps
    res[f"pr_delta_rmat{args.pr_scale}"] = {
        "iterations": st.iterations, "pull": sum(1 for i in range(n.value) if mode[i] == 0),
        "push_as_masked_pull": sum(1 for i in range(n.value) if mode[i] == 3),
        "push_with_atomics": sum(1 for i in range(n.value) if mode[i] == 1), "solve_ms": st.solve_ms,
        "ms_per_iter": st.solve_ms / st.iterations, "last_l1": st.last_error, "model_GBps": b / st.solve_ms / 1e6,
        "roofline_frac": b / st.solve_ms / 1e6 / HBM}
    L.gdn_pr_delta_plan_free(dplan)
    L.gdn_graph_free(go)
    L.gdn_graph_free(gi)
    del deg, scores, c

    # ---------------- SpMV (config 3)
    go, gi = build(args.spmv_scale, want_out=False)
    m, nnz = info(gi)
    for layout, lname in ((0, "csr"), (1, "pb")):
      for label, mk in (("const_0.2_0.3", lambda n, v: torch.full((n,), v, dtype=torch.float32, device=dev)),
                        ("uniform01", lambda n, v: torch.rand(n, dtype=torch.float32, device=dev))):
        Ax, x, y = mk(nnz, 0.2), mk(m, 0.3), torch.zeros(m, dtype=torch.float32, device=dev)
        plan = C.c_void_p()
        _cabi.check(L.gdn_spmv_plan_create(gi, ptr(Ax), layout, C.byref(plan)))
        b = int(L.gdn_spmv_bytes(plan))
        for _ in range(2):
            _cabi.check(L.gdn_spmv_dev(plan, ptr(Ax), ptr(x), ptr(y), None))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            _cabi.check(L.gdn_spmv_dev(plan, ptr(Ax), ptr(x), ptr(y), None))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        _cabi.check(L.gdn_spmv_plan_check(plan))
        L.gdn_spmv_plan_free(plan)
        res[f"spmv_rmat{args.spmv_scale}_{lname}_{label}"] = {
            "vertices": m, "nnz": nnz, "ms": dt * 1e3, "gflops": 2 * nnz / dt / 1e9, "algorithmic_GBps": b / dt / 1e9,
            "roofline_frac": b / dt / 1e9 / HBM, "algorithmic_bytes": b}
        del Ax, x, y
    L.gdn_graph_free(gi)

    # ---------------- BFS / SSSP / CC on one traversal graph
    go, gi = build(args.trav_scale)
    m, nnz = info(go)
    deg = torch.empty(m, dtype=torch.int32, device=dev)
    _cabi.check(L.gdn_graph_degrees_dev(go, ptr(deg), None))
    srcs = first_sources(deg)
    dist = torch.empty(m, dtype=torch.int32, device=dev)
    bplan = C.c_void_p()
    _cabi.check(L.gdn_bfs_plan_create(go, gi, 1, C.byref(bplan)))
    best = None
    for s in srcs:
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_bfs_run(bplan, int(s), ptr(dist), C.byref(st)))
        r = {"source": int(s), "ms": st.solve_ms, "levels": st.iterations, "edges_traversed": st.edges_traversed,
             "gteps": st.edges_traversed / st.solve_ms / 1e6}
        if best is None or r["gteps"] > best["gteps"]:
            best = r
    res[f"bfs_rmat{args.trav_scale}"] = best
    L.gdn_bfs_plan_free(bplan)
    for wname, w in (("unit", torch.ones(nnz, dtype=torch.int32, device=dev)),
                     ("u1_255", torch.randint(1, 256, (nnz,), dtype=torch.int32, device=dev))):
        splan = C.c_void_p()
        _cabi.check(L.gdn_sssp_plan_create(go, ptr(w), 1, C.byref(splan)))
        for delta in ((1,) if wname == "unit" else (16, 64)):
            st = _cabi.GdnStats()
            _cabi.check(L.gdn_sssp_dev(go, ptr(w), int(srcs[0]), delta, ptr(dist), C.byref(st)))
            res[f"sssp_rmat{args.trav_scale}_{wname}_delta{delta}_worklist"] = {
                "ms": st.solve_ms, "phases": st.iterations, "edges_traversed": st.edges_traversed,
                "gteps": st.edges_traversed / st.solve_ms / 1e6}
            st = _cabi.GdnStats()
            _cabi.check(L.gdn_sssp_run(splan, int(srcs[0]), delta, ptr(dist), C.byref(st)))
            res[f"sssp_rmat{args.trav_scale}_{wname}_delta{delta}_plan_dense"] = {
                "ms": st.solve_ms, "phases": st.iterations, "edges_traversed": st.edges_traversed,
                "gteps": st.edges_traversed / st.solve_ms / 1e6}
        L.gdn_sssp_plan_free(splan)
    del w
    # CC = weakly connected components of the directed graph: SV (out-CSR only, symmetric hook) and
    # Afforest (out- and in-CSR)
    comp = torch.empty(m, dtype=torch.int32, device=dev)
    for name, rev in (("sv_rounds", None), ("out_edges_only", None), ("afforest", gi)):
        # without the reverse graph: Afforest's sampling + one link pass over every out-edge (GDN_CC_SV=1: the SV rounds)
        if name == "sv_rounds":
            os.environ["GDN_CC_SV"] = "1"
        else:
            os.environ.pop("GDN_CC_SV", None)
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_cc_dev(go, rev, ptr(comp), C.byref(st)))
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_cc_dev(go, rev, ptr(comp), C.byref(st)))
        ncomp = int((comp == torch.arange(m, dtype=torch.int32, device=dev)).sum().item())
        res[f"cc_rmat{args.trav_scale}_{name}"] = {"ms": st.solve_ms, "rounds": st.iterations, "components": ncomp,
                                                    "gteps": nnz / st.solve_ms / 1e6}
    # BC from one source.  Byte model per reached edge: forward 4 (colidx) + 4 (depth probe) + 8 (path
    # count read-modify-write), backward 4 (colidx) + 16 (successor record); + 40 B per reached vertex
    sc = torch.zeros(m, dtype=torch.float32, device=dev)
    st = _cabi.GdnStats()
    _cabi.check(L.gdn_bc_dev(go, int(srcs[0]), ptr(sc), C.byref(st)))
    sc.zero_()
    st = _cabi.GdnStats()
    _cabi.check(L.gdn_bc_dev(go, int(srcs[0]), ptr(sc), C.byref(st)))
    reached_edges = st.edges_traversed // 2
    bc_bytes = 36 * reached_edges + 40 * int((sc == sc).sum().item())
    res[f"bc_rmat{args.trav_scale}"] = {"source": int(srcs[0]), "ms": st.solve_ms, "levels": st.iterations,
                                        "edge_visits": st.edges_traversed, "gteps": st.edges_traversed / st.solve_ms / 1e6,
                                        "model_GBps": bc_bytes / st.solve_ms / 1e6, "roofline_frac": bc_bytes / st.solve_ms / 1e6 / HBM}
    # the resident plan: BFS plan for the depths, heavy levels as propagation-blocked sweeps
    bplan = C.c_void_p()
    _cabi.check(L.gdn_bc_plan_create(go, gi, C.byref(bplan)))
    for _ in range(2):
        sc.zero_()
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_bc_run(bplan, int(srcs[0]), ptr(sc), C.byref(st)))
    res[f"bc_rmat{args.trav_scale}_plan"] = {"source": int(srcs[0]), "ms": st.solve_ms, "levels": st.iterations,
                                             "edge_visits": st.edges_traversed, "gteps": st.edges_traversed / st.solve_ms / 1e6,
                                             "prep_ms": st.prep_ms, "model_GBps": bc_bytes / st.solve_ms / 1e6,
                                             "roofline_frac": bc_bytes / st.solve_ms / 1e6 / HBM}
    L.gdn_bc_plan_free(bplan)
    del sc
    L.gdn_graph_free(gi)
    L.gdn_graph_free(go)
    del comp, dist, deg

    # ---------------- TC (config 4 stand-in): symmetrized on the device; both formulations of the count
    go, _ = build(args.tc_scale, want_in=False)
    h = C.c_void_p()
    _cabi.check(L.gdn_graph_symmetrize(go, C.byref(h)))
    L.gdn_graph_free(go)
    gm, gnnz = info(h)
    for form in ("auto", "u", "v"):
        if form == "auto":
            os.environ.pop("GDN_TC_FORM", None)
        else:
            os.environ["GDN_TC_FORM"] = form
        best = None
        for _ in range(3):
            total = C.c_uint64(0)
            st = _cabi.GdnStats()
            _cabi.check(L.gdn_tc_dev(h, 0, C.byref(total), C.byref(st)))
            if best is None or st.solve_ms < best[0]:
                best = (st.solve_ms, st.prep_ms, st.reserved)
        res[f"tc_rmat{args.tc_scale}_sym_{form}"] = {"vertices": gm, "sym_edges": gnnz, "dag_edges": st.edges_traversed,
                                                     "triangles": int(total.value), "count_ms": best[0], "prep_ms": best[1],
                                                     "form": "v-centric" if best[2] else "u-centric",
                                                     "gteps_dag": st.edges_traversed / best[0] / 1e6}
    os.environ.pop("GDN_TC_FORM", None)
    L.gdn_graph_free(h)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
