"""BASELINE.json's configs at THEIR size, against the CPU oracle (and the reference's own verifier where it is affordable):

  config 2  PageRank on an LJ-sized graph (soc-LiveJournal1 is not in the repository -- datasets/test.mk:5 is a wget
            line -- so R-MAT scale 22, 65 M edges, stands in; datasets/soc-LiveJournal1.mtx is used when present) to
            convergence: iteration count and L1 trace equal the oracle's, every score within 1e-4, PRVerifier's criterion
  config 3  SpMV fp32 on R-MAT scale 25 (529 M nonzeros), Ax and x ~ U(0,1): the resident plan AND the one-shot drop-in
            gdn_spmv against the oracle on EVERY row -- 1e-4 relative and SpmvVerifier's 5 sqrt(eps) criterion
  config 4  triangle count on symmetrized R-MAT scale 23 (129 M DAG edges; com-Orkut has 117 M) == the oracle's count
  config 5  PageRank on R-MAT scale 27 at N = 1: the headline plan (propagation-blocked, squished state) after 2 iterations
            against 2 oracle iterations on the full graph (same input on both sides each time), every one of the 134 M
            vertices within 1e-4 relative -- except hub rows of >= 10^4 in-edges, where the reference's one-by-one fp32
            sum drifts and the GPU value must be the one that matches an fp64 evaluation
            (N > 1: tests/test_gpu_multi.py and test_gpu_bench_sharded.py on one device)

Graphs are generated on the device (gdn_rmat_build, the generator the bench uses) and downloaded for the oracle."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from gardenia_amd import _cabi, graphio, solvers

pytestmark = pytest.mark.gpu

REFBIN = os.path.join(ROOT, "oracle", "_ref")


def _device_rmat(scale, want_out=True, want_in=True, symmetrize=False):
    """(out CSR, in CSR) as host arrays from the device generator."""
    L = _cabi.lib()
    go, gi = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi) if want_in else None))
    if symmetrize:
        gs = C.c_void_p()
        _cabi.check(L.gdn_graph_symmetrize(go, C.byref(gs)))
        L.gdn_graph_free(go)
        go = gs
    out = []
    for h, want in ((go, want_out), (gi, want_in)):
        if not want or not h:
            out.append(None)
            continue
        m, nnz = C.c_int32(), C.c_uint64()
        _cabi.check(L.gdn_graph_info(h, C.byref(m), C.byref(nnz), None, None))
        rp, ci = np.empty(m.value + 1, np.uint64), np.empty(nnz.value, np.int32)
        _cabi.check(L.gdn_graph_download(h, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p)))
        out.append(graphio.CSR(m.value, rp, ci))
    for h in (go, gi):
        if h:
            L.gdn_graph_free(h)
    return out


def test_config2_pagerank_lj_sized_to_convergence(orc):
    lj = os.path.join(ROOT, "datasets", "soc-LiveJournal1.mtx")
    if os.path.exists(lj):
        g = graphio.read_mtx(lj, False)
        gi = graphio.transpose(g)
    else:
        g, gi = _device_rmat(22)
    assert g.nnz > 60_000_000
    want, it, trace = orc.pr(gi, g.degrees())
    G = solvers.Graph(csr=g, in_csr=gi)
    scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st = solvers.PRSolver(G, scores)
    assert st["iterations"] == it
    np.testing.assert_allclose(st["trace"], trace, rtol=1e-3)  # sums of |new - old| near the stop: rounding noise of 3e-8
    rel = np.abs(scores - want) / want
    assert float(rel.max()) < 1e-4, float(rel.max())
    assert orc.pr_verify_error(g, scores) < 1e-4  # PRVerifier criterion, src/pr/verifier.cc:53
    # the merge-path layout (the other plan the drop-in can take) agrees as well
    os.environ["GDN_PR_LAYOUT"] = "csr"
    try:
        s2 = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        st2 = solvers.PRSolver(G, s2)
    finally:
        del os.environ["GDN_PR_LAYOUT"]
    assert st2["iterations"] == it and float((np.abs(s2 - want) / want).max()) < 1e-4
    # the reference's own verifier (src/pr/verifier.cc compiled in place) on the GPU scores
    if os.path.exists(os.path.join(REFBIN, "ref_pr")):
        with tempfile.TemporaryDirectory() as tmp:
            graphio.write_bin(os.path.join(tmp, "g"), g)
            scores.tofile(os.path.join(tmp, "pr"))
            p = subprocess.run([os.path.join(REFBIN, "ref_pr"), "verify", "bin", os.path.join(tmp, "g"), "0", "1",
                                os.path.join(tmp, "pr")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                               timeout=900, env=dict(os.environ, OMP_NUM_THREADS="16"))
            assert p.returncode == 0 and "Correct" in p.stdout, p.stdout[-1500:]


def test_config3_spmv_rmat25_every_row(orc):
    _, gi = _device_rmat(25, want_out=False)
    assert gi.m == 1 << 25 and gi.nnz > 500_000_000
    rng = np.random.default_rng(25)
    Ax = rng.random(gi.nnz, dtype=np.float32)
    x = rng.random(gi.m, dtype=np.float32)
    y0 = rng.random(gi.m, dtype=np.float32)
    want = orc.spmv(gi, Ax, x, y0)
    G = solvers.Graph(csr=gi, in_csr=gi)  # SpmvSolver multiplies the rows of in_rowptr / in_colidx
    tol = 5 * np.sqrt(np.finfo(np.float32).eps)  # src/spmv/verifier.cc:24
    # (1) the one-shot drop-in
    y = y0.copy()
    st = solvers.SpmvSolver(G, Ax, x, y)
    assert orc.spmv_max_rel_error(y, want) <= tol
    assert float((np.abs(y - want) / np.abs(want)).max()) < 1e-4
    assert st["edges_traversed"] == gi.nnz
    # (2) the resident plan (layout AUTO = propagation blocking with record tiers at this size), twice: y accumulates
    sp = solvers.ResidentSpMV(G, Ax)
    nh, nt, te = C.c_int32(0), C.c_int32(0), C.c_uint64(0)
    _cabi.check(_cabi.lib().gdn_spmv_plan_tiers(sp.plan, C.byref(nh), C.byref(nt), C.byref(te)))
    assert te.value > 0  # the headline layout, not the merge-path one
    y1 = sp.multiply(x, y0)
    assert orc.spmv_max_rel_error(y1, want) <= tol
    assert float((np.abs(y1 - want) / np.abs(want)).max()) < 1e-4
    y2 = sp.multiply(x, y1)
    want2 = orc.spmv(gi, Ax, x, want)
    assert orc.spmv_max_rel_error(y2, want2) <= tol
    sp.close()


def _device_standin(recipe, symmetrize=False, want_in=True):
    """A stand-in graph of graphio.LJ_LIKE / ORKUT_LIKE from the device generator (gdn_rmat_build_ex) as host arrays."""
    L = _cabi.lib()
    go, gi = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build_ex(recipe["scale"], recipe["n_edges"], *recipe["abc"], graphio.K_RAND_SEED, recipe["flags"],
                                    C.byref(go), C.byref(gi) if want_in else None))
    if symmetrize:
        gs = C.c_void_p()
        _cabi.check(L.gdn_graph_symmetrize(go, C.byref(gs)))
        L.gdn_graph_free(go)
        go = gs
    out = []
    for h in (go, gi):
        if not h:
            out.append(None)
            continue
        m, nnz = C.c_int32(), C.c_uint64()
        _cabi.check(L.gdn_graph_info(h, C.byref(m), C.byref(nnz), None, None))
        rp, ci = np.empty(m.value + 1, np.uint64), np.empty(nnz.value, np.int32)
        _cabi.check(L.gdn_graph_download(h, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p)))
        out.append(graphio.CSR(m.value, rp, ci))
        L.gdn_graph_free(h)
    return out


@pytest.mark.parametrize("abc,flags", [((0.57, 0.19, 0.19), 1), ((0.45, 0.22, 0.22), 3), ((0.5, 0.2, 0.2), 2), ((0.25, 0.25, 0.25), 3)])
def test_rmat_build_ex_equals_its_numpy_twin(abc, flags):
    """gdn_rmat_build_ex (quadrant probabilities, an edge count that is no multiple of 2^scale, ids without an edge dropped)
    == graphio.rmat_graph_ex bit for bit, both directions."""
    recipe = dict(scale=14, n_edges=(9 << 14) + 777, abc=abc, flags=flags)
    g, gi = _device_standin(recipe)
    want = graphio.rmat_graph_ex(14, recipe["n_edges"], abc, graphio.K_RAND_SEED, bool(flags & 1), bool(flags & 2))
    assert g.m == want.m and np.array_equal(g.rowptr, want.rowptr) and np.array_equal(g.colidx, want.colidx)
    wi = graphio.transpose(want)
    assert gi.m == want.m and np.array_equal(gi.rowptr, wi.rowptr) and np.array_equal(gi.colidx, wi.colidx)
    if flags & 2:
        assert ((g.degrees() + gi.degrees()) > 0).all() and g.m <= (1 << 14) and (g.m < (1 << 14) or abc[0] < 0.3)
    else:
        assert g.m == 1 << 14


@pytest.mark.parametrize("abc,permute,cuts", [((0.57, 0.19, 0.19), 1, (0, 5000, None, 11000, 1 << 14)), ((0.45, 0.22, 0.22), 0, (0, 1 << 13, 1 << 14)),
                                               ((0.57, 0.19, 0.19), 1, (0, 1 << 14))])
def test_rmat_build_range_gives_the_rows_of_the_whole_graph(abc, permute, cuts):
    """gdn_rmat_build_range (one rank's destination range of the generator's graph, built without the whole graph): for every
    range of a partition the in-CSR rows equal the whole in-CSR's rows, the out-degree contributions add up to the out-degree
    vector, and gdn_pr_squish_range + gdn_graph_pad_columns turn the range into the shard gdn_pr_squish_create +
    gdn_graph_slice_padded cut out of the whole graph (same bounds of the live-vertex space)."""
    import torch
    from gardenia_amd.sharded import padded_chunk
    L = _cabi.lib()
    scale, n_edges = 14, (9 << 14) + 777
    m = 1 << scale
    whole_out, whole_in = _device_standin(dict(scale=scale, n_edges=n_edges, abc=abc, flags=permute))
    if None in cuts:  # a range of ONE vertex, a live one (a range without a live vertex has no shard: gdn_graph_pad_columns refuses it)
        v = 5000 + int(np.nonzero((whole_in.degrees() + whole_out.degrees())[5000:] > 0)[0][0])
        cuts = (0, v, v + 1, 11000, 1 << 14)
    dev = torch.device("cuda", 0)
    out_deg = torch.zeros(m, dtype=torch.int32, device=dev)
    in_deg = torch.zeros(m, dtype=torch.int32, device=dev)
    world = len(cuts) - 1
    rows = []
    for r in range(world):
        h = C.c_void_p()
        _cabi.check(L.gdn_rmat_build_range(scale, n_edges, *abc, graphio.K_RAND_SEED, permute, cuts[r], cuts[r + 1], C.byref(h),
                                           C.c_void_p(out_deg.data_ptr())))
        mm, nnz = C.c_int32(), C.c_uint64()
        _cabi.check(L.gdn_graph_info(h, C.byref(mm), C.byref(nnz), None, None))
        assert mm.value == cuts[r + 1] - cuts[r]
        rp, ci = np.empty(mm.value + 1, np.uint64), np.empty(nnz.value, np.int32)
        _cabi.check(L.gdn_graph_download(h, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p)))
        lo, hi = int(whole_in.rowptr[cuts[r]]), int(whole_in.rowptr[cuts[r + 1]])
        assert np.array_equal(rp, whole_in.rowptr[cuts[r]:cuts[r + 1] + 1] - np.uint64(lo)) and np.array_equal(ci, whole_in.colidx[lo:hi])
        _cabi.check(L.gdn_graph_degrees_dev(h, C.c_void_p(in_deg[cuts[r]:cuts[r + 1]].data_ptr()), None))
        rows.append(h)
    torch.cuda.synchronize()
    assert np.array_equal(out_deg.cpu().numpy(), whole_out.degrees()) and np.array_equal(in_deg.cpu().numpy(), whole_in.degrees())
    # the relabelled, padded shard of every range == the one cut out of the whole squished graph
    g_in = C.c_void_p()
    _cabi.check(L.gdn_graph_upload(m, whole_in.nnz, whole_in.rowptr.ctypes.data_as(C.c_void_p), whole_in.colidx.ctypes.data_as(C.c_void_p), C.byref(g_in)))
    sq, gp, ms = C.c_void_p(), C.c_void_p(), C.c_int32()
    _cabi.check(L.gdn_pr_squish_create(g_in, C.c_void_p(out_deg.data_ptr()), C.byref(sq)))
    _cabi.check(L.gdn_pr_squish_info(sq, None, C.byref(ms), C.byref(gp), None))
    rb, sb = (C.c_int32 * (world + 1))(*cuts), (C.c_int32 * (world + 1))()
    for r in range(world):
        _cabi.check(L.gdn_pr_squish_range(rows[r], cuts[r], C.c_void_p(in_deg.data_ptr()), C.c_void_p(out_deg.data_ptr()), m, world + 1, rb, sb))
    live = ((whole_in.degrees() > 0) | (whole_out.degrees() > 0)).astype(np.int64)
    assert list(sb) == [int(live[:c].sum()) for c in cuts] and sb[world] == ms.value
    chunk = padded_chunk(list(sb))
    for r in range(world):
        _cabi.check(L.gdn_graph_pad_columns(rows[r], world, sb, chunk))
        want = C.c_void_p()
        _cabi.check(L.gdn_graph_slice_padded(gp, world, sb, chunk, r, C.byref(want)))
        got = []
        for h in (rows[r], want):
            mm, nnz = C.c_int32(), C.c_uint64()
            _cabi.check(L.gdn_graph_info(h, C.byref(mm), C.byref(nnz), None, None))
            rp, ci = np.empty(mm.value + 1, np.uint64), np.empty(nnz.value, np.int32)
            _cabi.check(L.gdn_graph_download(h, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p)))
            got.append((mm.value, rp, ci))
            L.gdn_graph_free(h)
        assert got[0][0] == got[1][0] == sb[r + 1] - sb[r]
        assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][2], got[1][2])
    L.gdn_pr_squish_free(sq)
    L.gdn_graph_free(g_in)
    # compaction is a property of the whole graph: refused
    h = C.c_void_p()
    assert L.gdn_rmat_build_range(scale, n_edges, *abc, graphio.K_RAND_SEED, 3, 0, 10, C.byref(h), None) != 0


def test_config2_standin_lj_like_pagerank(orc):
    """BASELINE config 2 on the LJ-LIKE stand-in (graphio.LJ_LIKE: ~5.9 M vertices, ~70 M directed edges, no isolated vertex,
    max degree ~2 x 10^4 -- soc-LiveJournal1 has 4.85 M / 69 M / 2 x 10^4) to convergence against the oracle."""
    g, gi = _device_standin(graphio.LJ_LIKE)
    deg, indeg = g.degrees(), gi.degrees()
    assert 5_000_000 < g.m < 7_000_000 and 65_000_000 < g.nnz < 75_000_000
    assert ((deg + indeg) > 0).all() and 8_000 < int(indeg.max()) < 40_000
    want, it, trace = orc.pr(gi, deg)
    scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st = solvers.PRSolver(solvers.Graph(csr=g, in_csr=gi), scores)
    assert st["iterations"] == it
    np.testing.assert_allclose(st["trace"], trace, rtol=1e-3)
    rel = np.abs(scores - want) / want
    print("LJ-like: |V| %d |E| %d max in-degree %d; %d iterations, layout %s, max rel %.2e, prep %.1f ms solve %.1f ms"
          % (g.m, g.nnz, int(indeg.max()), it, st["layout"], float(rel.max()), st["prep_ms"], st["solve_ms"]))
    assert float(rel.max()) < 1e-4
    assert orc.pr_verify_error(g, scores) < 1e-4


def test_config4_standin_orkut_like_triangle_count(orc):
    """BASELINE config 4 on the ORKUT-LIKE stand-in (graphio.ORKUT_LIKE, symmetrized: ~3.9 M vertices, ~117 M undirected
    edges, max degree ~3.5 x 10^4 -- com-Orkut has 3.07 M / 117 M / 3.3 x 10^4) against the oracle's count."""
    gs, _ = _device_standin(graphio.ORKUT_LIKE, symmetrize=True, want_in=False)
    deg = gs.degrees()
    assert 3_000_000 < gs.m < 4_200_000 and 200_000_000 < gs.nnz < 260_000_000 and (deg > 0).all()
    dag = orc.tc_orient(gs)
    want = orc.tc(dag)
    total, st = solvers.TCSolver(solvers.Graph(csr=gs, in_csr=gs))
    print("Orkut-like: |V| %d undirected edges %d max degree %d; %d triangles, formulation %d core %d, prep %.1f ms solve %.1f ms"
          % (gs.m, gs.nnz // 2, int(deg.max()), want, st["reserved"] & 0xFF, st["reserved"] >> 8, st["prep_ms"], st["solve_ms"]))
    assert total == want > 0


def _orkut():
    """datasets/com-Orkut.{mtx | vertex.bin + edge.bin + meta.txt} when somebody has put it there (datasets/test.mk:8 is a wget
    line; the file is not in the repository), symmetrized like `tc_omp_base` loads it (bin/run-mining.sh:3-9)"""
    base = os.path.join(ROOT, "datasets", "com-Orkut")
    if os.path.exists(base + ".mtx"):
        return graphio.read_mtx(base + ".mtx", True)
    if os.path.exists(base + ".meta.txt"):
        return graphio.symmetrize(graphio.read_bin(base))
    return None


def test_config4_triangle_count_orkut_sized(orc):
    gs = _orkut()
    if gs is None:
        gs, _ = _device_rmat(23, want_in=False, symmetrize=True)
        assert gs.m == 1 << 23
    assert gs.nnz > 230_000_000  # com-Orkut: 234 M CSR entries
    dag = orc.tc_orient(gs)
    assert dag.nnz * 2 == gs.nnz
    want = orc.tc(dag)
    total, st = solvers.TCSolver(solvers.Graph(csr=gs, in_csr=gs))
    assert total == want > 0
    assert st["edges_traversed"] == dag.nnz
    assert st["reserved"] & 0xFF == 3 and st["reserved"] >> 8 in (0, 8192, 12288, 16384)  # the forward count; R-MAT: with its core
    if gs.m == 1 << 23:
        assert st["reserved"] >> 8 == 12288  # (8192 / 12288 / 16384 ranks from 2^19 / 2^23 / 2^24 vertices)
    total2, _ = solvers.TCSolver(solvers.Graph(csr=dag), oriented=True)  # the DAG handed over like `Graph g(prefix, USE_DAG)`
    assert total2 == want


def test_tc_core_is_taken_where_the_graph_is_skewed(orc, monkeypatch):
    """From 2^21 vertices on the forward count takes a core (of 8192 ranks at this size) -- if at least 1/64 of the rows reach those ranks
    twice.  A uniform random graph of that size keeps every row with the hash-set kernel (the core kernel would find an
    empty list: 0.35 ms of launch and grabs, profiles/sessions/r04_98.sh); GDN_TC_CORE forces one whatever the graph."""
    monkeypatch.setenv("GDN_TC_FORM", "f")
    m = 1 << 21
    rng = np.random.default_rng(2121)
    src = rng.integers(0, m, 6 * m)
    dst = rng.integers(0, m, 6 * m)
    g = graphio.symmetrize(graphio.build_csr(m, src, dst))
    want = orc.tc(orc.tc_orient(g))
    total, st = solvers.TCSolver(solvers.Graph(csr=g, symmetrize=True))
    assert total == want and st["reserved"] == 3
    monkeypatch.setenv("GDN_TC_CORE", "8192")
    total, st = solvers.TCSolver(solvers.Graph(csr=g, symmetrize=True))
    assert total == want and st["reserved"] >> 8 in (0, 8192)  # (0: not a single row with two neighbours among the top ranks)
    monkeypatch.delenv("GDN_TC_CORE")
    gs, _ = _device_rmat(21, want_in=False, symmetrize=True)  # skewed, same size: the core by default
    total, st = solvers.TCSolver(solvers.Graph(csr=gs, in_csr=gs))
    assert total == orc.tc(orc.tc_orient(gs)) and st["reserved"] == 3 | (8192 << 8)


def test_pagerank_summation_order_is_the_only_difference(orc, monkeypatch):
    """VERDICT r4 item 1b on R-MAT scale 24 (268 M edges) to convergence, three solves of the PRSolver drop-in against orc.pr:
      default               the blocked layout's exact sums: whatever lies beyond 1e-4 is a row of >= 10^4 in-edges (where the
                            reference's one-by-one fp32 sum drifts) -- counted and bounded by the measurement
      GDN_PR_SUM=reference, rows of >= 10^4 in-edges re-summed in the reference's order: NO row beyond 1e-4
      GDN_PR_SUM=reference, every row: the oracle's bits in all 16.7 M scores, its iteration count, its trace."""
    g, gi = _device_rmat(24)
    deg = g.degrees()
    want, it, trace = orc.pr(gi, deg)
    G = solvers.Graph(csr=g, in_csr=gi)
    indeg = np.diff(gi.rowptr.astype(np.int64))

    def solve():
        scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        st = solvers.PRSolver(G, scores)
        assert st["layout"] == "pb" and st["iterations"] == it
        np.testing.assert_allclose(st["trace"], trace, rtol=1e-3)
        rel = np.abs(scores - want) / want
        return scores, rel, np.nonzero(rel >= 1e-4)[0]

    scores, rel, off = solve()
    print("PR RMAT-24 converged (%d iterations): rows beyond 1e-4: %d, max rel %.3e, min in-degree of those rows %s; rows of >= 10^4 "
          "in-edges: %d" % (it, len(off), float(rel.max()), int(indeg[off].min()) if len(off) else None, int((indeg >= 10_000).sum())))
    # measured (round 5, session r05_01): 22 iterations, NO row beyond 1e-4 at convergence (max 1.7e-5) although 2 325 rows have
    # >= 10^4 in-edges -- the drift of the reference's sequential sum peaks in the first iteration (equal terms), not here
    assert len(off) <= 20 and float(rel.max()) <= 5e-4
    if len(off):
        assert int(indeg[off].min()) >= 10_000
    monkeypatch.setenv("GDN_PR_SUM", "reference")
    monkeypatch.setenv("GDN_PR_SUM_MIN_DEGREE", "10000")
    _, rel2, off2 = solve()
    print("   hub rows in the reference's order: rows beyond 1e-4: %d, max rel %.3e" % (len(off2), float(rel2.max())))
    assert len(off2) == 0
    monkeypatch.delenv("GDN_PR_SUM_MIN_DEGREE")
    scores3, rel3, off3 = solve()
    assert np.array_equal(scores3.view(np.uint32), want.view(np.uint32))


def test_config5_pagerank_rmat27_two_iterations_vs_oracle(orc, monkeypatch):
    torch = pytest.importorskip("torch")
    L = _cabi.lib()
    dev = torch.device("cuda", 0)
    p = lambda t: C.c_void_p(t.data_ptr())
    go, gi = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build(27, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
    m, nnz = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(gi, C.byref(m), C.byref(nnz), None, None))
    m, nnz = m.value, nnz.value
    deg = torch.empty(m, dtype=torch.int32, device=dev)
    _cabi.check(L.gdn_graph_degrees_dev(go, p(deg), None))
    L.gdn_graph_free(go)
    # ---- GPU: the headline plan (what bench.py times), two iterations, exported to all m vertices
    plan = C.c_void_p()
    _cabi.check(L.gdn_pr_plan_create(gi, p(deg), m, 0, _cabi.GDN_LAYOUT_PB_SQUISHED, C.byref(plan)))
    ms = C.c_int32()
    _cabi.check(L.gdn_pr_plan_state_size(plan, C.byref(ms)))
    assert ms.value < m
    state = torch.empty(ms.value, dtype=torch.float32, device=dev)
    c = [torch.zeros(ms.value + 4, dtype=torch.float32, device=dev) for _ in range(2)]
    diff = torch.zeros(1, dtype=torch.float64, device=dev)
    got = torch.empty(m, dtype=torch.float32, device=dev)

    def gpu_iteration(h_scores):
        """one pull iteration of the plan from the m-entry score vector h_scores -> (scores, L1 change)"""
        start = torch.from_numpy(h_scores).to(dev)
        _cabi.check(L.gdn_pr_import_dev(plan, p(start), p(state), 0.85, None))
        dead = C.c_double(0)
        _cabi.check(L.gdn_pr_import_diff(plan, C.byref(dead)))
        _cabi.check(L.gdn_pr_contrib_dev(plan, p(state), p(c[0]), None))
        _cabi.check(L.gdn_pr_pull_dev(plan, p(c[0]), p(state), p(c[1]), p(diff), 0.85, None))
        _cabi.check(L.gdn_pr_export_dev(plan, p(state), p(got), 0.85, None))
        torch.cuda.synchronize()
        return got.cpu().numpy(), float(diff.item()) + dead.value

    # ---- oracle inputs: the full in-CSR on the host
    h_rp, h_ci = np.empty(m + 1, np.uint64), np.empty(nnz, np.int32)
    _cabi.check(L.gdn_graph_download(gi, h_rp.ctypes.data_as(C.c_void_p), h_ci.ctypes.data_as(C.c_void_p)))
    h_deg = deg.cpu().numpy()
    g_in = graphio.CSR(m, h_rp, h_ci)
    indeg = np.diff(h_rp.astype(np.int64))
    dead_v = (h_deg == 0) & (indeg == 0)
    assert dead_v.sum() > m // 3
    cur = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
    n_hub_rows = 0
    inputs, wants = [], []
    for it in range(2):
        # the SAME input on both sides: iteration 2 starts from the oracle's iteration-1 scores
        gpu, gpu_err = gpu_iteration(cur)
        want, cpu_err = orc.pr_iterate(g_in, h_deg, cur.copy(), 1)  # src/pr/omp_base.cc:23-34, all rows
        rel = np.abs(gpu - want) / want
        off = np.nonzero(rel >= 1e-4)[0]
        # The one documented deviation: a row with very many in-edges.  The reference adds its contributions one by
        # one in fp32 and drifts; the plan accumulates them exactly (2^-62 fixed point).  Such rows must be hub rows, few, and
        # the GPU value must be the one that agrees with an fp64 evaluation.
        # (observed, round 4: 379 rows and up to 1.76e-3 in the iteration from 1/m -- a row of 10^5 .. 10^6 EQUAL terms is where a
        # sequential fp32 sum drifts most --, fewer and smaller in the second; bench.py prints its sample as `parity_note`)
        assert len(off) <= 500 and (len(off) == 0 or float(rel.max()) <= 2.5e-3), (it, len(off), float(rel.max()))
        if len(off):
            assert indeg[off].min() >= 10_000, (it, int(indeg[off].min()))
            with np.errstate(divide="ignore"):
                contrib64 = cur.astype(np.float64) / h_deg.astype(np.float64)
            base = np.float32((np.float32(1.0) - np.float32(0.85)) / np.float32(m))
            for r in off[np.argsort(-rel[off])][:32]:
                exact = float(base) + 0.85 * float(contrib64[h_ci[h_rp[r]:h_rp[r + 1]]].sum())
                assert abs(gpu[r] - exact) <= 2e-7 * exact, (it, int(r), float(gpu[r]), exact)
                assert abs(gpu[r] - exact) < abs(want[r] - exact)
            n_hub_rows += len(off)
        # vertices without any edge sit at the base score on both sides, bit for bit
        assert np.array_equal(gpu[dead_v], want[dead_v])
        assert abs(gpu_err - cpu_err) <= 1e-4 * cpu_err
        inputs.append(cur)
        wants.append((want, cpu_err))
        cur = want
    print("rows beyond 1e-4 of the sequential fp32 sum (all hub rows, GPU == fp64):", n_hub_rows)
    _cabi.check(L.gdn_pr_plan_check(plan))
    L.gdn_pr_plan_free(plan)
    # ---- the contract-exact mode: the same plan built under GDN_PR_SUM=reference with the rows of >= 10^4
    # in-edges re-summed in the reference's order behind every pull (csrc/gdn_seqsum.hpp).  In BOTH iterations -- the first one,
    # from 1/m, is where the 379 rows above live -- NO row lies beyond north_star's 1e-4, and every re-summed row has the
    # oracle's bits.
    monkeypatch.setenv("GDN_PR_SUM", "reference")
    monkeypatch.setenv("GDN_PR_SUM_MIN_DEGREE", "10000")
    plan = C.c_void_p()
    _cabi.check(L.gdn_pr_plan_create(gi, p(deg), m, 0, _cabi.GDN_LAYOUT_PB_SQUISHED, C.byref(plan)))
    rows_, longest_, entries_ = C.c_int32(0), C.c_int32(0), C.c_uint64(0)
    _cabi.check(L.gdn_pr_plan_refsum_info(plan, C.byref(rows_), C.byref(longest_), C.byref(entries_), None))
    hub = indeg >= 10_000
    assert rows_.value == int(hub.sum()) and longest_.value == int(indeg.max()) and entries_.value >= int(indeg[hub].sum())
    for it in range(2):
        gpu, gpu_err = gpu_iteration(inputs[it])
        want, cpu_err = wants[it]
        rel = np.abs(gpu - want) / want
        assert float(rel.max()) < 1e-4, (it, int((rel >= 1e-4).sum()), float(rel.max()))
        assert np.array_equal(gpu[hub].view(np.uint32), want[hub].view(np.uint32)), it
        assert abs(gpu_err - cpu_err) <= 1e-4 * cpu_err
    print("GDN_PR_SUM=reference on the %d rows of >= 10^4 in-edges (longest %d): no row beyond 1e-4 in either iteration, those rows bit-equal"
          % (rows_.value, longest_.value))
    _cabi.check(L.gdn_pr_plan_check(plan))
    L.gdn_pr_plan_free(plan)
    L.gdn_graph_free(gi)
