"""GPU parity tests: the HIP path, called through the C-ABI, against (a) the golden vectors
produced by the reference itself and (b) the CPU oracle on seeded inputs.  Bit-exact for BFS
depths, SSSP distances, CC labels and TC counts; PageRank/SpMV within 1e-4 relative
(BASELINE.json north_star)."""
import os

import numpy as np
import pytest

from conftest import csr_from, golden
from gardenia_amd import graphio, solvers

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4  # north_star: "PR/SpMV within 1e-4 relative"


def _graph(d, sym=False):
    g = csr_from(d)
    gi = csr_from(d, "in_") if "in_rowptr" in d else None
    return solvers.Graph(csr=g, in_csr=gi, symmetrize=sym, need_reverse=gi is not None and not sym)


# ------------------------------------------------------------------ BFS
@pytest.mark.parametrize("case", ["test_bc_dir", "test_bc_sym", "chesapeake_sym", "4_dir", "rmat10_dir", "rmat12_dir"])
@pytest.mark.parametrize("with_reverse", [True, False])
def test_bfs_golden(case, with_reverse):
    d = golden("bfs_" + case)
    g = solvers.Graph(csr=csr_from(d), in_csr=csr_from(d, "in_") if with_reverse else None)
    dist = np.full(g.V(), solvers.MYINFINITY, np.int32)
    st = solvers.BFSSolver(g, int(d["source"]), dist)
    assert np.array_equal(dist, d["dist"])
    assert st["iterations"] >= 1


@pytest.mark.parametrize("scale,ef,seed", [(14, 16, 1), (16, 16, 2), (18, 8, 3)])
def test_bfs_vs_oracle_rmat(orc, scale, ef, seed):
    g = graphio.rmat_graph(scale, ef, seed=seed)
    gi = graphio.transpose(g)
    s = graphio.first_nonisolated(g)
    want = orc.bfs_serial(g, s)
    for rev in (gi, None):
        dist = np.full(g.m, solvers.MYINFINITY, np.int32)
        st = solvers.BFSSolver(solvers.Graph(csr=g, in_csr=rev), s, dist)
        assert np.array_equal(dist, want)
        reached = want != solvers.MYINFINITY
        assert st["edges_traversed"] == int(g.degrees()[reached].astype(np.int64).sum())


@pytest.mark.parametrize("scale,ef,seed", [(12, 16, 4), (16, 16, 5), (19, 16, 6), (17, 64, 7)])
def test_bfs_resident_dense_levels(orc, scale, ef, seed):
    """gdn_bfs_plan_*: heavy levels run as propagation-blocked sweeps over all in-edges."""
    g = graphio.rmat_graph(scale, ef, seed=seed)
    G = solvers.Graph(csr=g, need_reverse=True)
    bfs = solvers.ResidentBFS(G, dense=True)
    deg = g.degrees()
    sources = [graphio.first_nonisolated(g), int(np.argmax(deg)), int(np.nonzero(deg)[0][-1])]
    for s in sources:
        dist, st = bfs.run(s)
        want = orc.bfs_serial(g, s)
        assert np.array_equal(dist, want), (s, int((dist != want).sum()))
        assert st["edges_traversed"] == int(deg[want != solvers.MYINFINITY].astype(np.int64).sum())
    bfs.close()


@pytest.mark.parametrize("scale,ef,seed,hub_min,outer", [(19, 16, 8, None, None), (19, 48, 9, "0", "0"), (20, 8, 10, "100000", None),
                                                         (19, 16, 8, "0", "1"), (20, 8, 10, "100000", "1"), (22, 8, 12, None, "1")])
def test_bfs_bottom_up_heads_vs_oracle(orc, monkeypatch, capfd, scale, ef, seed, hub_min, outer):
    """The bottom-up step's HEAD records (bfs_hub_head_kernel: every row's in-neighbour of highest out-degree, as a hub index
    tested against LDS bits or as a vertex id tested against the frontier bitmap, + the row's out-degree) exist from 2^24
    edges on; forced here at sizes the serial oracle walks, with the hub test always on / gated off / at its default: depths
    and the traversed-edge count equal the oracle's from hub, leaf and late sources, and the trace shows the heads at work.
    `outer` = GDN_BFS_HUBS2: the 2^21 highest out-degrees named by RANK and tested against the rank-indexed frontier bits
    (default from 2^27 vertices on): below 2^21 vertices every head is a rank, at scale 22 ranks and vertex ids mix."""
    if outer is not None:
        monkeypatch.setenv("GDN_BFS_HUBS2", outer)
        monkeypatch.setenv("GDN_BFS_DEFER_DEPTH", outer)  # ... whose levels leave the distances to one pass at the end of the search
        monkeypatch.setenv("GDN_BFS_TD_DEFER_MIN", "1")   # (top-down levels behind a snapshot too, whatever their weight)
        monkeypatch.setenv("GDN_BFS_REC_COMPACT", outer)  # ... and the wave kernel on the compact copy of the records (default from 2^25 vertices on)
    monkeypatch.setenv("GDN_BFS_HEADS_MIN_NNZ", "1")
    monkeypatch.setenv("GDN_BFS_TRACE", "1")
    if hub_min is not None:
        monkeypatch.setenv("GDN_BFS_HUB_MIN", hub_min)
    g = graphio.rmat_graph(scale, ef, seed=seed)
    G = solvers.Graph(csr=g, need_reverse=True)
    capfd.readouterr()
    bfs = solvers.ResidentBFS(G, dense=True)
    assert "heads of the bottom-up step" in capfd.readouterr().err
    deg = g.degrees()
    sources = [graphio.first_nonisolated(g), int(np.argmax(deg)), int(np.nonzero(deg)[0][-1]), int(np.nonzero(deg == 1)[0][0])]
    by_head = 0
    for s in sources:
        dist, st = bfs.run(s)
        log = capfd.readouterr().err
        import re
        by_head += sum(int(x) for x in re.findall(r"(\d+) rows by their hub head", log))
        want = orc.bfs_serial(g, s)
        assert np.array_equal(dist, want), (s, int((dist != want).sum()))
        assert st["edges_traversed"] == int(deg[want != solvers.MYINFINITY].astype(np.int64).sum())
    bfs.close()
    assert by_head > 0


def test_bfs_deferred_depths_beyond_the_pool_of_kept_levels(orc, monkeypatch, capfd):
    """Deferred depths keep at most BFS_DEFER_MAX = 8 level bitmaps per search; the levels after that write their depths as
    before.  A sparse uniform random graph (2^20 vertices, average out-degree 2.5: ~20 levels, a dozen of them with a snapshot
    in front) with every snapshot level deferred: the pool runs out, the trace says 8 kept levels, and the depths -- from kept
    levels, from levels that wrote directly behind them and from the light ones -- equal the oracle's."""
    for k, v in (("GDN_BFS_HEADS_MIN_NNZ", "1"), ("GDN_BFS_HUB_MIN", "0"), ("GDN_BFS_DEFER_DEPTH", "1"), ("GDN_BFS_TD_DEFER_MIN", "1"),
                 ("GDN_BFS_REC_COMPACT", "1"), ("GDN_BFS_TRACE", "1")):
        monkeypatch.setenv(k, v)
    m, src, dst = graphio.uniform_edges(1 << 20, 5 << 19, seed=77)
    g = graphio.build_csr(m, src, dst)
    G = solvers.Graph(csr=g, need_reverse=True)
    bfs = solvers.ResidentBFS(G, dense=True)
    capfd.readouterr()
    s = graphio.first_nonisolated(g)
    dist, st = bfs.run(s)
    log = capfd.readouterr().err
    import re
    kept = [int(x) for x in re.findall(r"distances of (\d+) kept levels", log)]
    want = orc.bfs_serial(g, s)
    assert np.array_equal(dist, want), int((dist != want).sum())
    assert kept == [8] and int(want[want != solvers.MYINFINITY].max()) >= 12, (kept, log[-1500:])
    bfs.close()


@pytest.mark.parametrize("blind", ["0", "1"])
def test_bfs_top_down_without_the_read_in_front_of_the_atomic(orc, monkeypatch, blind):
    """Top-down levels claim a vertex with an atomic OR on the visited bitmap; the plain read in front of it is skipped (`blind`)
    in the early levels of a graph without hubs (default rule: not skewed and < 1/8 of the rows visited).  Forced on / off here
    on an R-MAT graph (hubs: thousands of edges race for one word) and a uniform one: depths and traversed edges as the oracle's."""
    monkeypatch.setenv("GDN_BFS_TD_BLIND", blind)
    monkeypatch.setenv("GDN_BFS_HEADS_MIN_NNZ", "1")
    m, src, dst = graphio.uniform_edges(1 << 16, 1 << 20, seed=5)
    for g in (graphio.rmat_graph(15, 16, seed=9), graphio.build_csr(m, src, dst)):
        G = solvers.Graph(csr=g, need_reverse=True)
        bfs = solvers.ResidentBFS(G, dense=True)
        deg = g.degrees()
        for s in (graphio.first_nonisolated(g), int(np.argmax(deg))):
            dist, st = bfs.run(s)
            want = orc.bfs_serial(g, s)
            assert np.array_equal(dist, want), (s, int((dist != want).sum()))
            assert st["edges_traversed"] == int(deg[want != solvers.MYINFINITY].astype(np.int64).sum())
        bfs.close()


def test_bfs_star_and_chain(orc):
    # a hub with 20000 out-neighbours (big-row path) feeding a chain (many tiny levels)
    n = 20001 + 300
    src = np.concatenate([np.zeros(20000, np.int64), np.arange(20000, n - 1)])
    dst = np.concatenate([np.arange(1, 20001), np.arange(20001, n)])
    g = graphio.build_csr(n, src, dst)
    dist = np.full(n, solvers.MYINFINITY, np.int32)
    solvers.BFSSolver(solvers.Graph(csr=g, in_csr=graphio.transpose(g)), 0, dist)
    assert np.array_equal(dist, orc.bfs_serial(g, 0))
    assert dist[-1] == 301


@pytest.mark.parametrize("coop", [None, "0", "1"])
def test_bfs_cooperative_light_levels(orc, monkeypatch, coop):
    """High-diameter graphs: light levels that outgrow the one-workgroup kernel run on a cooperative grid with a barrier
    per level (bfs_td_coop_kernel; GDN_BFS_COOP=1: from the first light level, 0: never, default: after 8 light levels in
    a row).  A 300 x 300 lattice (598 levels of up to 300 vertices), a lattice with random shortcuts, and an R-MAT graph
    (where the path must not change anything): depths exact, from the resident plan and the drop-in."""
    if coop is not None:
        monkeypatch.setenv("GDN_BFS_COOP", coop)
    rng = np.random.default_rng(8)
    m, src, dst = graphio.grid2d_edges(300, 300)
    extra = rng.integers(0, m, (200, 2))
    graphs = [graphio.build_csr(m, src, dst),
              graphio.build_csr(m, np.concatenate([src, extra[:, 0]]), np.concatenate([dst, extra[:, 1]])),
              graphio.rmat_graph(15, 16, seed=9)]
    for g in graphs:
        G = solvers.Graph(csr=g, need_reverse=True)
        bfs = solvers.ResidentBFS(G, dense=True)
        for s in (graphio.first_nonisolated(g), int(g.m // 2 + 7)):
            want = orc.bfs_serial(g, s)
            dist, st = bfs.run(s)
            assert np.array_equal(dist, want), (s, int((dist != want).sum()))
            assert st["edges_traversed"] == int(g.degrees()[want != solvers.MYINFINITY].astype(np.int64).sum())
            d2 = np.full(g.m, solvers.MYINFINITY, np.int32)
            solvers.BFSSolver(solvers.Graph(csr=g), s, d2)
            assert np.array_equal(d2, want)
        bfs.close()


# ------------------------------------------------------------------ PR
@pytest.fixture(params=["csr", "pb", "fused"])
def pr_layout(request, monkeypatch):
    """Both edge layouts of the PageRank plan (include/gardenia_hip.h GDN_LAYOUT_*) under the per-iteration loop, and the
    whole solve in one cooperative launch (pr_fused_kernel, what gdn_pr picks by itself below 2^22 edges)."""
    if request.param == "fused":
        monkeypatch.setenv("GDN_PR_FUSED", "1")
        monkeypatch.delenv("GDN_PR_LAYOUT", raising=False)
    else:
        monkeypatch.setenv("GDN_PR_LAYOUT", request.param)
    return request.param


@pytest.mark.parametrize("case", ["test_pr", "chesapeake_sym", "rmat10", "rmat12"])
def test_pr_golden(case, pr_layout):
    d = golden("pr_" + case)
    g = solvers.Graph(csr=csr_from(d), in_csr=csr_from(d, "in_"))
    scores = np.full(g.V(), np.float32(1.0) / np.float32(g.V()), np.float32)
    st = solvers.PRSolver(g, scores)
    assert st["iterations"] == int(d["iterations"])
    np.testing.assert_allclose(scores, d["scores"], rtol=REL_TOL, atol=0)
    assert abs(st["last_error"] - d["trace"][-1]) < 5e-7


def test_pr_trace_of_reference_repo():
    import json, os
    from conftest import GOLDEN
    gold = json.load(open(os.path.join(GOLDEN, "pr_trace_golden.json")))
    g = solvers.Graph(os.path.join(GOLDEN, "graphs", "test_pr"), "mtx", False, True)
    scores = np.full(g.V(), np.float32(1.0) / np.float32(g.V()), np.float32)
    st = solvers.PRSolver(g, scores)
    assert st["iterations"] == gold["iterations"]
    assert "%.6f" % st["last_error"] == "%.6f" % gold["trace"][-1]


@pytest.mark.parametrize("scale,ef,seed", [(14, 16, 5), (17, 16, 6), (12, 64, 7)])
def test_pr_vs_oracle_rmat(orc, scale, ef, seed, pr_layout):
    g = graphio.rmat_graph(scale, ef, seed=seed)
    gi = graphio.transpose(g)
    want, it, trace = orc.pr(gi, g.degrees())
    scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st = solvers.PRSolver(solvers.Graph(csr=g, in_csr=gi), scores)
    assert st["iterations"] == it
    np.testing.assert_allclose(scores, want, rtol=REL_TOL, atol=0)
    assert orc.pr_verify_error(g, scores) < 1e-4  # PRVerifier criterion, src/pr/verifier.cc:53


@pytest.mark.parametrize("layout", ["csr", "pb", "pb_unsquished", "pb_tiers"])
def test_pr_reference_sum_mode_has_the_reference_bits(orc, layout, monkeypatch):
    """GDN_PR_SUM=reference (diagnostic): behind every pull the rows are summed again the way src/pr/omp_base.cc:27-33 sums
    them -- one fp32 add per in-edge in CSR order.  The solve then has the oracle's BITS in every score, whatever layout the
    plan streams: the order of the additions is the only thing in which the library's PageRank differs from the
    reference's.  With GDN_PR_SUM_MIN_DEGREE only the rows with that many in-edges are re-summed."""
    g = graphio.rmat_graph(15, 16, seed=44)
    gi = graphio.transpose(g)
    want, it, trace = orc.pr(gi, g.degrees())
    G = solvers.Graph(csr=g, in_csr=gi)
    monkeypatch.setenv("GDN_PR_LAYOUT", "csr" if layout == "csr" else "pb")
    if layout == "pb_unsquished":
        monkeypatch.setenv("GDN_PR_SQUISH", "0")
    if layout == "pb_tiers":
        monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1")  # record tiers on a graph this small
    plain = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st0 = solvers.PRSolver(G, plain)
    assert st0["iterations"] == it
    monkeypatch.setenv("GDN_PR_SUM", "reference")
    scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st = solvers.PRSolver(G, scores)
    assert st["layout"] == ("csr" if layout == "csr" else "pb")
    assert st["iterations"] == it
    assert np.array_equal(scores.view(np.uint32), want.view(np.uint32))  # bit for bit
    np.testing.assert_allclose(st["trace"], trace, rtol=1e-9)  # (double sums of the same terms in another order)
    # only the rows of >= 64 in-edges: those rows carry sequential sums (of inputs that differ in their last bits by now),
    # the others the plan's own -- everything within 1e-4, and the run differs from the plain one
    monkeypatch.setenv("GDN_PR_SUM_MIN_DEGREE", "64")
    part = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st2 = solvers.PRSolver(G, part)
    assert st2["iterations"] == it
    np.testing.assert_allclose(part, want, rtol=REL_TOL, atol=0)
    indeg = np.diff(gi.rowptr.astype(np.int64))
    small = indeg < 64
    assert float((np.abs(part[small] - plain[small]) / plain[small]).max()) < 1e-5
    monkeypatch.delenv("GDN_PR_SUM")
    again = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    solvers.PRSolver(G, again)
    assert np.array_equal(again, plain)  # the mode leaves nothing behind


@pytest.mark.parametrize("wg_min", ["1", "4000", "4000000000"])
def test_pr_reference_order_sums_of_adversarial_values(monkeypatch, wg_min):
    """The kernels behind GDN_PR_SUM=reference (the stage pass: contributions through LDS slices into row order; the scans:
    gdn_seqsum.hpp's parity functions instead of a chain of additions) on contributions chosen to hit every branch of
    its arithmetic -- equal terms, exact ties, powers of two, zeros, denormals, terms above the running sum; then a negative
    term and an infinity (the hardware path) --, on rows of 1 ... 9 000 in-edges, the long ones on a wave or on a workgroup
    each: every score has the bits of
    base + damping * (the contributions added one by one in fp32, in CSR order), src/pr/omp_base.cc:27-33."""
    import ctypes as C
    from gardenia_amd import _cabi
    from test_seqsum_math import _case
    L = _cabi.lib()
    rng = np.random.default_rng(len(wg_min))
    m = 9000
    rows = [np.arange(1, m), rng.choice(m, 5000, replace=False), rng.choice(m, 700, replace=False), np.arange(0, m, 7)]
    rows += [rng.choice(m, int(d), replace=False) for d in rng.integers(0, 40, m - len(rows))]
    dst = np.concatenate([np.full(len(r), i) for i, r in enumerate(rows)])
    src = np.concatenate(rows)
    g = graphio.build_csr(m, src.astype(np.int64), dst.astype(np.int64))  # out-CSR of src -> dst
    gi = graphio.transpose(g)
    monkeypatch.setenv("GDN_PR_SUM", "reference")
    # rows of at least wg_min in-edges are summed by a WORKGROUP each (16 waves chain the pairs of 16 blocks per round:
    # pr_refseg_wg_kernel), the others by a wave each -- "1": every row longer than one block; "4000000000": none
    monkeypatch.setenv("GDN_PR_SUM_WG_MIN", wg_min)
    h, plan = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_graph_upload(m, gi.nnz, gi.rowptr.ctypes.data_as(C.c_void_p), gi.colidx.ctypes.data_as(C.c_void_p), C.byref(h)))

    def dev(a):
        p = C.c_void_p()
        _cabi.check(L.gdn_dev_alloc(a.nbytes, C.byref(p)))
        _cabi.check(L.gdn_dev_upload(p, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return p

    def host(p, n, dt):
        a = np.empty(n, dt)
        _cabi.check(L.gdn_dev_download(a.ctypes.data_as(C.c_void_p), p, a.nbytes))
        return a

    deg = dev(np.maximum(g.degrees(), 1).astype(np.int32))
    _cabi.check(L.gdn_pr_plan_create(h, deg, m, 0, 0, C.byref(plan)))  # merge-path layout: any float may be a contribution
    base = np.float32((np.float32(1.0) - np.float32(0.85)) / np.float32(m))
    for variant in ("finite", "weird"):
        contrib = np.concatenate([_case(k, m // 9, rng)[:m // 9] for k in range(9)] + [np.zeros(m, np.float32)])[:m].astype(np.float32)
        contrib = contrib[rng.permutation(m)]
        if variant == "weird":
            contrib[rng.integers(0, m, 3)] *= np.float32(-1.0)
            contrib[int(rng.integers(0, m))] = np.inf
        sc, c0, c1, diff = dev(np.full(m, 1.0 / m, np.float32)), dev(contrib), dev(np.zeros(m, np.float32)), dev(np.zeros(1, np.float64))
        _cabi.check(L.gdn_pr_pull_dev(plan, c0, sc, c1, diff, 0.85, None))
        got = host(sc, m, np.float32)
        want = np.empty(m, np.float32)
        with np.errstate(all="ignore"):
            for r in range(m):
                t = np.float32(0)
                for v in contrib[gi.colidx[int(gi.rowptr[r]):int(gi.rowptr[r + 1])]]:
                    t = np.float32(t + v)
                want[r] = np.float32(base + np.float32(np.float32(0.85) * t))
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (variant, np.nonzero(~same)[0][:5], got[~same][:5], want[~same][:5])
        for p in (sc, c0, c1, diff):
            L.gdn_dev_free(p)
    L.gdn_pr_plan_free(plan)
    L.gdn_graph_free(h)
    L.gdn_dev_free(deg)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("layout", [0, 1])
def test_pr_sharded_data_path_on_one_device(orc, world, layout):
    """Every C-ABI call of the multi-GPU path (row slices, plans with row_base and m_local < m_global,
    contrib slices at their global place) on one device, all-gather emulated by copies: the result must
    equal the single-plan solve."""
    g = graphio.rmat_graph(15, 16, seed=31)
    m = g.m - 5  # not divisible by the world size
    src, dst = graphio.csr_to_coo(g)
    keep = (src < m) & (dst < m)
    g = graphio.build_csr(m, src[keep], dst[keep])
    gi = graphio.transpose(g)
    want, it, trace = orc.pr(gi, g.degrees())
    sh = solvers.ResidentPageRankShards(solvers.Graph(csr=g, in_csr=gi), world, layout)
    scores, it2, err = sh.solve()
    sh.close()
    assert it2 == it
    np.testing.assert_allclose(scores, want, rtol=REL_TOL, atol=0)
    assert abs(err - trace[-1]) < 1e-6


@pytest.mark.parametrize("world,parts", [(2, 4), (3, 3), (8, 4), (1, 5)])
def test_pr_row_range_parts_equal_whole_iterations(world, parts):
    """gdn_pr_pull_rows_dev: an iteration issued as row-range parts (what the multi-GPU pipeline does) gives the
    same bits as gdn_pr_pull_dev on the propagation-blocked layout (integer accumulation, same bins)."""
    g = graphio.rmat_graph(16, 16, seed=33)
    m = g.m - 7
    src, dst = graphio.csr_to_coo(g)
    keep = (src < m) & (dst < m)
    g = graphio.build_csr(m, src[keep], dst[keep])
    G = solvers.Graph(csr=g, need_reverse=True)
    res = []
    for p in (1, parts):
        sh = solvers.ResidentPageRankShards(G, world, 1, parts=p)
        res.append(sh.solve())
        sh.close()
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert np.array_equal(res[0][0], res[1][0])


def test_pr_row_range_part_contract():
    """After part j every row below its row_end is final in scores and contrib_out (the rows the pipeline sends)."""
    import ctypes as C
    from gardenia_amd import _cabi
    L = _cabi.lib()
    g = graphio.rmat_graph(17, 16, seed=34)
    gi = graphio.transpose(g)
    m = g.m
    h, plan = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_graph_upload(m, gi.nnz, gi.rowptr.ctypes.data_as(C.c_void_p), gi.colidx.ctypes.data_as(C.c_void_p),
                                   C.byref(h)))

    def dev(a):
        p = C.c_void_p()
        _cabi.check(L.gdn_dev_alloc(a.nbytes, C.byref(p)))
        _cabi.check(L.gdn_dev_upload(p, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return p

    def host(p, n, dt):
        a = np.empty(n, dt)
        _cabi.check(L.gdn_dev_download(a.ctypes.data_as(C.c_void_p), p, a.nbytes))
        return a

    deg = dev(g.degrees().astype(np.int32))
    _cabi.check(L.gdn_pr_plan_create(h, deg, m, 0, 1, C.byref(plan)))
    init = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
    zeros = np.zeros(m, np.float32)
    # reference: one whole iteration
    sc, c0, c1, diff = dev(init), dev(zeros), dev(zeros), dev(np.zeros(1, np.float64))
    _cabi.check(L.gdn_pr_contrib_dev(plan, sc, c0, None))
    _cabi.check(L.gdn_pr_pull_dev(plan, c0, sc, c1, diff, 0.85, None))
    want_s, want_c, want_d = host(sc, m, np.float32), host(c1, m, np.float32), host(diff, 1, np.float64)[0]
    # the same iteration in 5 uneven parts
    sc2, c2 = dev(init), dev(zeros)
    cuts = [0, 1000, 1004, m // 3, m - 12345, m]
    for j in range(len(cuts) - 1):
        flags = (1 if j == 0 else 0) | (2 if j == len(cuts) - 2 else 0)
        _cabi.check(L.gdn_pr_pull_rows_dev(plan, c0, sc2, c2, diff, 0.85, cuts[j], cuts[j + 1], flags, None))
        r1 = cuts[j + 1]
        assert np.array_equal(host(sc2, m, np.float32)[:r1], want_s[:r1]), j
        got_c = host(c2, m, np.float32)[:r1]
        assert np.array_equal(got_c, want_c[:r1], equal_nan=True), j
    assert host(diff, 1, np.float64)[0] == want_d
    _cabi.check(L.gdn_pr_plan_check(plan))
    L.gdn_pr_plan_free(plan)
    L.gdn_graph_free(h)
    for p in (deg, sc, c0, c1, diff, sc2, c2):
        L.gdn_dev_free(p)


def test_pr_plan_takes_the_block_reserved_at_process_start(orc, capfd, monkeypatch):
    """gdn_dev_reserve (measurement hook of DESIGN 4.1): a block set aside before any build becomes the per-iteration scratch
    array of the first blocked PageRank plan that fits into it -- same scores; a block that is too small is left alone."""
    import ctypes as C
    from gardenia_amd import _cabi
    L = _cabi.lib()
    g = graphio.rmat_graph(15, 16, seed=45)
    gi = graphio.transpose(g)
    want, it, _ = orc.pr(gi, g.degrees())
    G = solvers.Graph(csr=g, in_csr=gi)
    monkeypatch.setenv("GDN_PR_LAYOUT", "pb")
    monkeypatch.setenv("GDN_PR_PLACE_TRACE", "1")
    for nbytes, taken in ((64, False), (64 << 20, True)):
        _cabi.check(L.gdn_dev_reserve(nbytes))
        scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        st = solvers.PRSolver(G, scores)
        assert st["iterations"] == it
        np.testing.assert_allclose(scores, want, rtol=REL_TOL, atol=0)
        assert ("reserved at process start" in capfd.readouterr().err) == taken
    _cabi.check(L.gdn_dev_reserve(0))


@pytest.mark.parametrize("layout", [0, 1])
def test_pr_ticketed_parts_contract(layout):
    """gdn_pr_pull_parts_dev + gdn_pr_wait_part_dev: ONE launch per phase whose rows become final part by part.  A copy queued
    on a second stream BEHIND part j's waiter (what the sharded driver does with its all-gather) sees scores and contrib_out
    final for every row below row_end[j]; scores, contributions and the L1 change have the bits of the whole-iteration pull;
    a second iteration reuses the counters (targets accumulate).  Blocked layout and merge-path layout (all parts at once)."""
    import ctypes as C
    import torch
    from gardenia_amd import _cabi
    L = _cabi.lib()
    g = graphio.rmat_graph(17, 16, seed=35)
    gi = graphio.transpose(g)
    m = g.m
    h, plan = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_graph_upload(m, gi.nnz, gi.rowptr.ctypes.data_as(C.c_void_p), gi.colidx.ctypes.data_as(C.c_void_p), C.byref(h)))
    dev = torch.device("cuda", 0)
    deg = torch.from_numpy(g.degrees().astype(np.int32)).to(dev)
    pp = lambda t: C.c_void_p(t.data_ptr())
    _cabi.check(L.gdn_pr_plan_create(h, pp(deg), m, 0, layout, C.byref(plan)))
    init = torch.full((m,), 1.0 / m, dtype=torch.float32, device=dev)

    def fresh():
        return init.clone(), [torch.zeros(m, dtype=torch.float32, device=dev) for _ in range(2)], torch.zeros(1, dtype=torch.float64, device=dev)

    # reference: two whole iterations
    sc, cc, dd = fresh()
    _cabi.check(L.gdn_pr_contrib_dev(plan, pp(sc), pp(cc[0]), None))
    want = []
    for k in range(2):
        _cabi.check(L.gdn_pr_pull_dev(plan, pp(cc[k & 1]), pp(sc), pp(cc[(k + 1) & 1]), pp(dd), 0.85, None))
        torch.cuda.synchronize()
        want.append((sc.clone(), cc[(k + 1) & 1].clone(), float(dd.item())))
    # the same two iterations ticketed, 5 uneven parts; behind every waiter a copy of the part's rows on the side stream
    sc2, c2, d2 = fresh()
    _cabi.check(L.gdn_pr_contrib_dev(plan, pp(sc2), pp(c2[0]), None))
    ends = [1000, 1004, m // 3, m - 12345, m]
    arr = (C.c_int32 * len(ends))(*ends)
    side = torch.cuda.Stream(device=dev)
    for k in range(2):
        torch.cuda.synchronize()
        _cabi.check(L.gdn_pr_pull_parts_dev(plan, pp(c2[k & 1]), pp(sc2), pp(c2[(k + 1) & 1]), pp(d2), 0.85, len(ends), arr, None))
        snaps = []
        with torch.cuda.stream(side):
            for j, r1 in enumerate(ends):
                _cabi.check(L.gdn_pr_wait_part_dev(plan, j, C.c_void_p(side.cuda_stream)))
                snaps.append((sc2[:r1].clone(), c2[(k + 1) & 1][:r1].clone()))
        torch.cuda.synchronize()
        for j, r1 in enumerate(ends):
            assert torch.equal(snaps[j][0], want[k][0][:r1]), (k, j)
            assert torch.equal(snaps[j][1].view(torch.int32), want[k][1][:r1].view(torch.int32)), (k, j)
        assert float(d2.item()) == want[k][2]
    _cabi.check(L.gdn_pr_plan_check(plan))
    with pytest.raises(RuntimeError):  # no such part
        _cabi.check(L.gdn_pr_wait_part_dev(plan, len(ends), None))
    L.gdn_pr_plan_free(plan)
    L.gdn_graph_free(h)


@pytest.mark.parametrize("world,parts", [(1, 1), (2, 3)])
def test_pr_hub_row_tier(orc, monkeypatch, world, parts):
    """Hub-ROW tier (phase A sums the edges into the highest in-degree rows in LDS, one partial per chunk and row;
    optional, GDN_PB_HUB_ROWS=1): only built with full-size chunks, so the chunk size is forced here; bit-identical to
    the plan without it."""
    monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1000")
    monkeypatch.setenv("GDN_PB_HUB_MIN", "1")
    monkeypatch.setenv("GDN_PB_LOG_CHUNK", "15")
    g = graphio.rmat_graph(17, 16, seed=37)
    gi = graphio.transpose(g)
    want, it, trace = orc.pr(gi, g.degrees())
    G = solvers.Graph(csr=g, in_csr=gi)
    res = []
    for rows in ("0", "1"):
        monkeypatch.setenv("GDN_PB_HUB_ROWS", rows)
        sh = solvers.ResidentPageRankShards(G, world, 1, parts=parts)
        res.append(sh.solve())
        sh.close()
    assert res[0][1] == res[1][1] == it
    assert np.array_equal(res[0][0], res[1][0])
    np.testing.assert_allclose(res[1][0], want, rtol=REL_TOL, atol=0)


@pytest.mark.parametrize("world,parts", [(1, 1), (2, 4), (3, 2)])
def test_pr_hub_tier_on_row_shards(orc, monkeypatch, world, parts):
    """Hub tier + vertex-range shards (m_local < m_global, row_base) + row-range parts: the multi-GPU configuration of
    bench.py at a size the oracle solves; the tier is forced on for small shards."""
    monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1000")
    monkeypatch.setenv("GDN_PB_HUB_MIN", "1")
    g = graphio.rmat_graph(16, 16, seed=36)
    m = g.m - 11
    src, dst = graphio.csr_to_coo(g)
    keep = (src < m) & (dst < m)
    g = graphio.build_csr(m, src[keep], dst[keep])
    gi = graphio.transpose(g)
    want, it, trace = orc.pr(gi, g.degrees())
    sh = solvers.ResidentPageRankShards(solvers.Graph(csr=g, in_csr=gi), world, 1, parts=parts)
    import ctypes as C
    from gardenia_amd import _cabi
    nh = C.c_int32(0)
    _cabi.check(_cabi.lib().gdn_pr_plan_hubs(sh.ranks[0]["plan"], C.byref(nh), None))
    assert nh.value > 0, "the hub tier was not built"
    scores, it2, err = sh.solve()
    sh.close()
    assert it2 == it
    np.testing.assert_allclose(scores, want, rtol=REL_TOL, atol=0)
    assert abs(err - trace[-1]) < 1e-6


TIER_ENV = {"GDN_PB_HUB_MIN_NNZ": "1000", "GDN_PB_HUB_MIN": "64", "GDN_PB_MID_CAP": "1500"}


@pytest.mark.parametrize("world,parts", [(1, 1), (2, 3)])
def test_pr_mid_tiers_are_bitwise_neutral(orc, monkeypatch, world, parts):
    """Record tiers of the PB layout (hubs + two mid tiers: their edges skip phase A and are read by phase B as 32-bit
    (source, row) records, values from per-iteration tables of fixed-point codes), forced on at a size the oracle
    solves (GDN_PB_MID_CAP bounds a tier's sources so that two tiers form): bit-identical to the plan without mid
    tiers and to the plan without any tier, on whole graphs and on row shards issued in parts."""
    import ctypes as C
    from gardenia_amd import _cabi
    g = graphio.rmat_graph(17, 16, seed=41)
    gi = graphio.transpose(g)
    want, it, trace = orc.pr(gi, g.degrees())
    G = solvers.Graph(csr=g, in_csr=gi)
    for k, v in TIER_ENV.items():
        monkeypatch.setenv(k, v)
    res = []
    # (il: the mid tiers' record streams in lane-interleaved blocks of 256 -- phase B form 2, the default -- or plain)
    for hubs, mid, il in (("0", "0", "1"), ("1", "0", "1"), ("1", "2", "1"), ("1", "2", "0")):
        monkeypatch.setenv("GDN_PB_HUBS", hubs)
        monkeypatch.setenv("GDN_PB_MID", mid)
        monkeypatch.setenv("GDN_PB_REC_IL", il)
        sh = solvers.ResidentPageRankShards(G, world, 1, parts=parts)
        nh, nt, ns, ne = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_uint64(0)
        _cabi.check(_cabi.lib().gdn_pr_plan_hubs(sh.ranks[0]["plan"], C.byref(nh), None))
        _cabi.check(_cabi.lib().gdn_pr_plan_mid(sh.ranks[0]["plan"], C.byref(nt), C.byref(ns), C.byref(ne)))
        if hubs == "0":
            assert nh.value == 0 and nt.value == 0
        elif mid == "0":
            assert nh.value > 0 and nt.value == 0 and ne.value == 0
        else:
            assert nh.value > 0 and nt.value == 2 and ns.value > 1500 and ne.value > 0, (nh.value, nt.value, ns.value)
        scores, it2, err = sh.solve()
        sh.close()
        res.append((scores, it2, err))
    assert res[0][1] == res[1][1] == res[2][1] == res[3][1] == it
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][0], res[2][0]) and np.array_equal(res[0][0], res[3][0])
    assert res[0][2] == res[1][2] == res[2][2] == res[3][2]
    np.testing.assert_allclose(res[2][0], want, rtol=REL_TOL, atol=0)
    assert abs(res[2][2] - trace[-1]) < 1e-6


@pytest.mark.parametrize("world,parts", [(1, 1), (2, 3)])
def test_pr_placement_search_is_bitwise_neutral(orc, monkeypatch, capfd, world, parts):
    """gdn_pr_plan_create times fresh allocations of a blocked plan's streamed arrays and keeps the fastest ones
    (PbPlacer, DESIGN 4.1: from 2^28 edges on; forced here on every array of a small plan, tiers on).  A result must not
    depend on where an array lives: same bits, iteration count and final error as the plan left where hipMalloc put it,
    on whole graphs and on row shards; and gdn_pr_plan_move (the measurement hook) moves a live plan's arrays between
    two solves without changing a bit."""
    from gardenia_amd import _cabi
    g = graphio.rmat_graph(17, 16, seed=43)
    gi = graphio.transpose(g)
    want, it, trace = orc.pr(gi, g.degrees())
    G = solvers.Graph(csr=g, in_csr=gi)
    for k, v in TIER_ENV.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("GDN_PLACE_MIN_EDGES", "1")
    monkeypatch.setenv("GDN_PLACE_MIN_BYTES", "1")
    monkeypatch.setenv("GDN_PR_PLACE_TRACE", "1")
    monkeypatch.setenv("GDN_PR_PLACE_COPIES", "1")  # (the copies of phase B's streams are searched on request only since round 5)
    res = []
    for tries in ("0", "2", "hook"):
        monkeypatch.setenv("GDN_PR_PLACE", "0" if tries == "hook" else tries)
        capfd.readouterr()
        sh = solvers.ResidentPageRankShards(G, world, 1, parts=parts)
        log = capfd.readouterr().err
        assert ("[pr place]" in log) == (tries == "2"), log[-400:]
        if tries == "2":
            assert "[pr place] %-12s fresh" % "vals" in log  # (scratch of one iteration: candidates are fresh allocations)
            for name in ("V", "mid records", "hub records", "U"):
                assert "[pr place] %-12s try" % name in log, name
        if tries == "hook":  # every array group of the finished plan into a fresh allocation
            for r in sh.ranks:
                if r["plan"]:
                    _cabi.check(_cabi.lib().gdn_pr_plan_move(r["plan"], 127))
        res.append(sh.solve())
        sh.close()
    for scores, it2, err in res:
        assert it2 == it and np.array_equal(scores, res[0][0]) and err == res[0][2]
    np.testing.assert_allclose(res[0][0], want, rtol=REL_TOL, atol=0)


def test_spmv_placement_search_is_bitwise_neutral(orc, monkeypatch, capfd):
    """The same search in gdn_spmv_plan_create (blocked layout with tiers, forced on a small matrix)."""
    monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1")
    monkeypatch.setenv("GDN_PLACE_MIN_EDGES", "1")
    monkeypatch.setenv("GDN_PLACE_MIN_BYTES", "1")
    monkeypatch.setenv("GDN_SPMV_PLACE_TRACE", "1")
    g = graphio.rmat_graph(16, 16, seed=47)
    gi = graphio.transpose(g)
    rng = np.random.default_rng(47)
    Ax, x = rng.random(gi.nnz, dtype=np.float32), rng.random(gi.m, dtype=np.float32)
    y0 = rng.random(gi.m, dtype=np.float32)
    got = []
    for tries in ("0", "2"):
        monkeypatch.setenv("GDN_SPMV_PLACE", tries)
        capfd.readouterr()
        sp = solvers.ResidentSpMV(solvers.Graph(csr=g, in_csr=gi), Ax, layout=1)
        log = capfd.readouterr().err
        assert ("[spmv place]" in log) == (tries != "0"), log[-400:]
        got.append(sp.multiply(x, y0))
        sp.close()
    assert np.array_equal(got[0], got[1])
    want = orc.spmv(gi, Ax, x, y0)
    np.testing.assert_allclose(got[1], want, rtol=REL_TOL, atol=0)


@pytest.mark.parametrize("tiers", [False, True])
def test_pr_squished_vertex_space_is_bitwise_neutral(orc, monkeypatch, tiers):
    """GDN_LAYOUT_PB_SQUISHED: the plan's per-iteration state leaves out the vertices without any edge (they keep the base
    score).  Host API with and without it: same bits, same iteration count, same last error; and arbitrary (non-uniform)
    start scores, so that the L1 change of the left-out vertices in the first iteration matters."""
    import ctypes as C
    from gardenia_amd import _cabi
    if tiers:
        for k, v in TIER_ENV.items():
            monkeypatch.setenv(k, v)
    monkeypatch.setenv("GDN_PR_LAYOUT", "pb")
    g = graphio.rmat_graph(17, 4, seed=52)  # sparse: more than half of the vertices have no edge at all
    gi = graphio.transpose(g)
    G = solvers.Graph(csr=g, in_csr=gi)
    live = (g.degrees() > 0) | (gi.degrees() > 0)
    assert 0.2 < live.mean() < 0.8
    want, it, trace = orc.pr(gi, g.degrees())
    res = []
    for sq in ("0", "1"):
        monkeypatch.setenv("GDN_PR_SQUISH", sq)
        scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        st = solvers.PRSolver(G, scores)
        res.append((scores, st))
    assert np.array_equal(res[0][0], res[1][0])
    assert res[0][1]["iterations"] == res[1][1]["iterations"] == it
    assert abs(res[0][1]["last_error"] - res[1][1]["last_error"]) < 1e-12
    np.testing.assert_allclose(res[1][0], want, rtol=REL_TOL, atol=0)
    # resident calls with non-uniform start scores: first-iteration L1 change incl. the left-out vertices
    L = _cabi.lib()
    h = C.c_void_p()
    _cabi.check(L.gdn_graph_upload(g.m, gi.nnz, gi.rowptr.ctypes.data_as(C.c_void_p), gi.colidx.ctypes.data_as(C.c_void_p),
                                   C.byref(h)))

    def dev(a):
        p = C.c_void_p()
        _cabi.check(L.gdn_dev_alloc(max(a.nbytes, 4), C.byref(p)))
        _cabi.check(L.gdn_dev_upload(p, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return p

    rng = np.random.default_rng(3)
    start = rng.random(g.m).astype(np.float32)
    start /= np.float32(start.sum())
    deg = dev(g.degrees().astype(np.int32))
    outs = []
    for layout in (_cabi.GDN_LAYOUT_PB, _cabi.GDN_LAYOUT_PB_SQUISHED):
        plan = C.c_void_p()
        _cabi.check(L.gdn_pr_plan_create(h, deg, g.m, 0, layout, C.byref(plan)))
        ms = C.c_int32(0)
        _cabi.check(L.gdn_pr_plan_state_size(plan, C.byref(ms)))
        assert ms.value == (g.m if layout == _cabi.GDN_LAYOUT_PB else int(live.sum()))
        full = dev(start)
        state = dev(np.zeros(ms.value, np.float32))
        c = [dev(np.zeros(ms.value, np.float32)), dev(np.zeros(ms.value, np.float32))]
        diff = dev(np.zeros(1, np.float64))
        _cabi.check(L.gdn_pr_import_dev(plan, full, state, 0.85, None))
        dead = C.c_double(0)
        _cabi.check(L.gdn_pr_import_diff(plan, C.byref(dead)))
        _cabi.check(L.gdn_pr_contrib_dev(plan, state, c[0], None))
        diffs = []
        for k in range(3):
            _cabi.check(L.gdn_pr_pull_dev(plan, c[k & 1], state, c[(k + 1) & 1], diff, 0.85, None))
            d = np.empty(1, np.float64)
            _cabi.check(L.gdn_dev_download(d.ctypes.data_as(C.c_void_p), diff, 8))
            diffs.append(d[0] + (dead.value if k == 0 else 0.0))
        _cabi.check(L.gdn_pr_export_dev(plan, state, full, 0.85, None))
        got = np.empty(g.m, np.float32)
        _cabi.check(L.gdn_dev_download(got.ctypes.data_as(C.c_void_p), full, 4 * g.m))
        _cabi.check(L.gdn_pr_plan_check(plan))
        outs.append((got, diffs))
        L.gdn_pr_plan_free(plan)
        for p_ in (full, state, c[0], c[1], diff):
            L.gdn_dev_free(p_)
    L.gdn_dev_free(deg)
    L.gdn_graph_free(h)
    assert np.array_equal(outs[0][0], outs[1][0])
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-12, atol=0)
    base = np.float32((np.float32(1.0) - np.float32(0.85)) / np.float32(g.m))
    assert np.all(outs[1][0][~live] == base)


def test_pr_pb_rejects_out_of_range_scores(monkeypatch):
    """The fixed-point codes are made once per source (phase A's slice, the tier tables): a contribution outside [0,1]
    must still raise GDN_ERR_OVERFLOW, whichever tier its source sits in."""
    for k, v in TIER_ENV.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("GDN_PR_LAYOUT", "pb")
    g = graphio.rmat_graph(15, 16, seed=5)
    G = solvers.Graph(csr=g, need_reverse=True)
    deg = g.degrees()
    for victim in (int(np.argmax(deg)), int(np.flatnonzero(deg == 1)[0])):  # a hub source, a main-layout source
        scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        scores[victim] = np.float32(3.0) * max(int(deg[victim]), 1)  # contribution 3.0
        with pytest.raises(Exception) as ei:
            solvers.PRSolver(G, scores)
        assert "fixed-point" in str(ei.value) or "OVERFLOW" in str(ei.value).upper(), str(ei.value)


@pytest.mark.parametrize("layout_env", [{"GDN_PB_V8": "1"}, {"GDN_PB_V8": "1", "GDN_PB_HUB_MIN_NNZ": "1000", "GDN_PB_HUB_MIN": "1"}])
def test_pr_delta_coded_rows_are_bitwise_neutral(orc, monkeypatch, layout_env):
    """Optional 8-bit delta coding of the row stream (PbPlan::v8, off by default because its decode costs more than
    the bytes it saves): filler edges keep every distance <= 255; the result equals the u16 layout bit for bit.
    The graph has long empty row ranges, so fillers are really needed."""
    rng = np.random.default_rng(12)
    m = 1 << 16
    # sparse rows: ~2 edges per 1000 rows inside a bin -> distances far above 255, plus a few dense rows
    src = rng.integers(0, m, 6000)
    dst = rng.integers(0, m, 6000)
    src = np.concatenate([src, rng.integers(0, m, 20000)])
    dst = np.concatenate([dst, np.repeat(rng.integers(0, m, 20), 1000)])
    g = graphio.build_csr(m, src, dst)
    G = solvers.Graph(csr=g, need_reverse=True)
    monkeypatch.setenv("GDN_PR_LAYOUT", "pb")
    ref = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
    st0 = solvers.PRSolver(G, ref)
    for k, v in layout_env.items():
        monkeypatch.setenv(k, v)
    got = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
    st1 = solvers.PRSolver(G, got)
    assert st0["iterations"] == st1["iterations"]
    assert np.array_equal(ref, got)
    want, it, _ = orc.pr(graphio.transpose(g), g.degrees())
    assert it == st1["iterations"]
    np.testing.assert_allclose(got, want, rtol=REL_TOL, atol=0)


@pytest.mark.parametrize("env", [{}, {"GDN_PB_HUB_MIN_NNZ": "1000", "GDN_PB_HUB_MIN": "1"},
                                 {"GDN_PB_HUB_MIN_NNZ": "1000", "GDN_PB_HUB_MIN": "1", "GDN_PB_V8": "1"},
                                 {"GDN_PB_COMPACT": "0"}, TIER_ENV,
                                 dict(TIER_ENV, GDN_PB_REC_IL="0", GDN_PB_V_IL="0")])  # (the last: plain, not lane-interleaved, streams)
def test_spmv_pb_layout_variants(orc, monkeypatch, env):
    """SpMV on the propagation-blocked layout: compacted (default), with the hub tier forced on at this size (hub edges
    multiply x[hub] by their own Ax inside phase B), with delta-coded rows, and uncompacted -- all against the oracle,
    and bit-identical to each other (integer accumulation)."""
    g = graphio.rmat_graph(16, 16, seed=14)
    rng = np.random.default_rng(4)
    Ax = (rng.random(g.nnz) - 0.5).astype(np.float32)
    x = (rng.random(g.m) - 0.5).astype(np.float32)
    y0 = rng.random(g.m).astype(np.float32)
    G = solvers.Graph(csr=g, in_csr=g)
    want = orc.spmv(g, Ax, x, y0)
    monkeypatch.setenv("GDN_PB_HUBS", "0")
    sp = solvers.ResidentSpMV(G, Ax, layout=1)
    ref = sp.multiply(x, y0)
    sp.close()
    monkeypatch.delenv("GDN_PB_HUBS")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sp = solvers.ResidentSpMV(G, Ax, layout=1)
    if "GDN_PB_MID_CAP" in env:  # hubs + two mid tiers really are in use
        import ctypes as C
        from gardenia_amd import _cabi
        nh, nt, ne = C.c_int32(0), C.c_int32(0), C.c_uint64(0)
        _cabi.check(_cabi.lib().gdn_spmv_plan_tiers(sp.plan, C.byref(nh), C.byref(nt), C.byref(ne)))
        assert nh.value > 0 and nt.value == 2 and 0 < ne.value < g.nnz, (nh.value, nt.value, ne.value)
    got = sp.multiply(x, y0)
    got2 = sp.multiply(x, got)  # a second multiply on the same plan: y accumulates
    sp.close()
    assert np.array_equal(got, ref)
    np.testing.assert_allclose(got, want, rtol=REL_TOL, atol=1e-6)
    np.testing.assert_allclose(got2, orc.spmv(g, Ax, x, want), rtol=REL_TOL, atol=1e-5)
    assert orc.spmv_max_rel_error(got, want) <= 5 * np.sqrt(np.finfo(np.float32).eps)


def test_spmv_delta_coded_rows(orc, monkeypatch):
    g = graphio.rmat_graph(15, 8, seed=13)
    rng = np.random.default_rng(3)
    Ax = (rng.random(g.nnz) - 0.5).astype(np.float32)
    x = (rng.random(g.m) - 0.5).astype(np.float32)
    y0 = rng.random(g.m).astype(np.float32)
    G = solvers.Graph(csr=g, in_csr=g)
    want = orc.spmv(g, Ax, x, y0)
    ys = []
    for v8 in ("0", "1"):
        monkeypatch.setenv("GDN_PB_V8", v8)
        sp = solvers.ResidentSpMV(G, Ax, layout=1)
        ys.append(sp.multiply(x, y0))
        sp.close()
    assert np.array_equal(ys[0], ys[1])
    np.testing.assert_allclose(ys[1], want, rtol=REL_TOL, atol=1e-6)


@pytest.mark.parametrize("shape", ["star_out", "star_in", "two_hubs_only", "empty_rows"])
def test_pr_hub_tier_degenerate_graphs(orc, monkeypatch, shape):
    """Corner cases of the two-layout plan: every edge leaves a hub (the main layout is EMPTY), every edge enters one
    row, no edges at all in most bins."""
    monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1")
    monkeypatch.setenv("GDN_PB_HUB_MIN", "1")
    monkeypatch.setenv("GDN_PR_LAYOUT", "pb")
    m = 40000
    if shape == "star_out":  # vertex 0 -> everybody
        src, dst = np.zeros(m - 1, np.int64), np.arange(1, m)
    elif shape == "star_in":  # everybody -> vertex 0
        src, dst = np.arange(1, m), np.zeros(m - 1, np.int64)
    elif shape == "two_hubs_only":  # two sources own all edges, scattered rows
        rng = np.random.default_rng(1)
        dst = rng.permutation(m)[:5000]
        src = np.where(np.arange(5000) % 2 == 0, 7, 31000)
    else:  # a few edges between ids far apart
        src, dst = np.array([5, 39999, 20000]), np.array([39999, 5, 1])
    g = graphio.build_csr(m, src, dst)
    gi = graphio.transpose(g)
    want, it, _ = orc.pr(gi, g.degrees())
    got = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
    st = solvers.PRSolver(solvers.Graph(csr=g, in_csr=gi), got)
    assert st["iterations"] == it
    np.testing.assert_allclose(got, want, rtol=REL_TOL, atol=0)


def test_pr_hub_tier_is_bitwise_neutral(monkeypatch):
    """PB layout with the hub tier (edges of the highest-degree sources bypass the per-edge value stream) against the
    same layout without it: integer accumulation makes the two bit-identical; and the tier really is in use."""
    import ctypes as C
    from gardenia_amd import _cabi
    L = _cabi.lib()
    g = graphio.rmat_graph(20, 24, seed=35)  # > 2^24 edges: the tier is only built for large graphs
    gi = graphio.transpose(g)
    m = g.m
    h = C.c_void_p()
    _cabi.check(L.gdn_graph_upload(m, gi.nnz, gi.rowptr.ctypes.data_as(C.c_void_p), gi.colidx.ctypes.data_as(C.c_void_p),
                                   C.byref(h)))

    def dev(a):
        p = C.c_void_p()
        _cabi.check(L.gdn_dev_alloc(a.nbytes, C.byref(p)))
        _cabi.check(L.gdn_dev_upload(p, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return p

    deg = dev(g.degrees().astype(np.int32))
    out = []
    for hubs in ("0", "1"):
        monkeypatch.setenv("GDN_PB_HUBS", hubs)
        plan = C.c_void_p()
        _cabi.check(L.gdn_pr_plan_create(h, deg, m, 0, 1, C.byref(plan)))
        nh, he = C.c_int32(0), C.c_uint64(0)
        _cabi.check(L.gdn_pr_plan_hubs(plan, C.byref(nh), C.byref(he)))
        if hubs == "0":
            assert nh.value == 0 and he.value == 0
        else:
            assert 0 < nh.value <= 32768 and 0 < he.value < gi.nnz
            hub_edges = he.value
        sc = dev(np.full(m, np.float32(1.0) / np.float32(m), np.float32))
        c = [dev(np.zeros(m, np.float32)), dev(np.zeros(m, np.float32))]
        diff = dev(np.zeros(1, np.float64))
        _cabi.check(L.gdn_pr_contrib_dev(plan, sc, c[0], None))
        diffs = []
        for it in range(4):
            _cabi.check(L.gdn_pr_pull_dev(plan, c[it & 1], sc, c[(it + 1) & 1], diff, 0.85, None))
            d = np.empty(1, np.float64)
            _cabi.check(L.gdn_dev_download(d.ctypes.data_as(C.c_void_p), diff, 8))
            diffs.append(d[0])
        _cabi.check(L.gdn_pr_plan_check(plan))
        s = np.empty(m, np.float32)
        _cabi.check(L.gdn_dev_download(s.ctypes.data_as(C.c_void_p), sc, 4 * m))
        out.append((s, diffs))
        L.gdn_pr_plan_free(plan)
        for p in (sc, c[0], c[1], diff):
            L.gdn_dev_free(p)
    L.gdn_dev_free(deg)
    L.gdn_graph_free(h)
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    # the hubs are the top sources: their share of the edges is far above their share of the vertices
    assert hub_edges > gi.nnz // 50


def test_pr_is_bitwise_reproducible(monkeypatch):
    monkeypatch.setenv("GDN_PR_LAYOUT", "csr")  # the merge-path layout is the reproducible one
    g = graphio.rmat_graph(15, 16, seed=9)
    G = solvers.Graph(csr=g, need_reverse=True)
    runs = []
    for _ in range(3):
        s = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        solvers.PRSolver(G, s)
        runs.append(s)
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])


def test_pr_interleaved_streams_are_bitwise_neutral(orc, monkeypatch):
    """The lane-interleaved forms of phase B's streams at a size where they are on by default -- V in blocks of 512 edges
    (bins start on multiples of 512 from ~2^14 main-layout edges per bin on), the mid tiers' records in blocks of 256 --
    against the plain forms: same bits, same iteration count, and within 1e-4 of the oracle."""
    g = graphio.rmat_graph(20, 16, seed=5)
    gi = graphio.transpose(g)
    want, it, _ = orc.pr(gi, g.degrees())
    G = solvers.Graph(csr=g, in_csr=gi)
    monkeypatch.setenv("GDN_PR_LAYOUT", "p")
    monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1")
    res = []
    for v_il, rec_il in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("GDN_PB_V_IL", v_il)
        monkeypatch.setenv("GDN_PB_REC_IL", rec_il)
        s = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        st = solvers.PRSolver(G, s)
        res.append((s, st["iterations"]))
    assert all(r[1] == it for r in res)
    assert all(np.array_equal(res[0][0], r[0]) for r in res[1:])
    np.testing.assert_allclose(res[0][0], want, rtol=REL_TOL, atol=0)


def test_pr_hub_row_spanning_tiles(orc, pr_layout):
    # vertex 0 has 50000 in-neighbours: its row spans > 12 merge-path tiles.  The reference adds
    # such a row sequentially in fp32 (omp_base.cc:28-29), which by itself drifts ~6e-4 from the
    # exact sum; the tile-wise sum here is closer to exact.  So: rows of ordinary length must
    # match the oracle to 1e-4, and EVERY row must match an fp64-accumulated evaluation of the
    # same five iterations to 1e-4 (the oracle's hub row does not).
    n = 60000
    src = np.concatenate([np.arange(1, 50001), np.zeros(100, np.int64), np.arange(1, 2000)])
    dst = np.concatenate([np.zeros(50000, np.int64), np.arange(50001, 50101), np.arange(2, 2001)])
    g = graphio.build_csr(n, src, dst)
    gi = graphio.transpose(g)
    deg = g.degrees()
    want = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
    orc.pr_iterate(gi, deg, want, 5)
    scores = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
    solvers.PRSolver(solvers.Graph(csr=g, in_csr=gi), scores, epsilon=0.0, max_iter=5)
    # fp64 evaluation of the same Jacobi iteration
    s64 = np.full(n, 1.0 / n)
    isrc, idst = graphio.csr_to_coo(gi)  # (row = dst, col = src)
    with np.errstate(divide="ignore"):
        for _ in range(5):
            c = s64 / deg
            s64 = (1.0 - 0.85) / n + 0.85 * np.bincount(isrc, weights=c[idst], minlength=n)
    if pr_layout == "csr":
        np.testing.assert_allclose(scores, s64, rtol=REL_TOL, atol=0)
        assert abs(want[0] - s64[0]) > abs(scores[0] - s64[0])  # the tile-wise sum is the more exact one
    else:
        # the PB layout accumulates a row in LDS one term at a time, like the reference's
        # sequential loop: every row must be within 1e-4 of the oracle OR of the fp64 evaluation
        ok = (np.abs(scores - want) <= REL_TOL * np.abs(want)) | (np.abs(scores - s64) <= REL_TOL * np.abs(s64))
        assert ok.all()
    indeg = gi.degrees()
    short = indeg < 1000
    fed_by_hub = np.zeros(n, bool)
    fed_by_hub[50001:50101] = True
    ok = short & ~fed_by_hub
    np.testing.assert_allclose(scores[ok], want[ok], rtol=REL_TOL, atol=0)


def test_pr_pb_layout_odd_sizes(orc, monkeypatch):
    """PB layout with m not a multiple of the block size, empty chunks/bins and a dense hub."""
    monkeypatch.setenv("GDN_PR_LAYOUT", "pb")
    rng = np.random.default_rng(3)
    n = 70001
    src = np.concatenate([rng.integers(0, n, 400000), rng.integers(60000, n, 5000), np.full(3000, 17)])
    dst = np.concatenate([rng.integers(0, 2000, 400000), rng.integers(0, n, 5000), rng.integers(0, n, 3000)])
    g = graphio.build_csr(n, src, dst)
    gi = graphio.transpose(g)
    want, it, _ = orc.pr(gi, g.degrees())
    scores = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
    st = solvers.PRSolver(solvers.Graph(csr=g, in_csr=gi), scores)
    assert st["iterations"] == it
    np.testing.assert_allclose(scores, want, rtol=REL_TOL, atol=0)


# ------------------------------------------------------------------ SpMV
@pytest.mark.parametrize("case", ["test_bc", "chesapeake_sym", "rmat10_rand"])
def test_spmv_golden(orc, case):
    d = golden("spmv_" + case)
    gi = csr_from(d, "in_")
    if "Ax" in d:
        Ax, x, y0 = d["Ax"], d["x"], d["y0"]
    else:
        Ax, x, y0 = np.full(gi.nnz, 0.2, np.float32), np.full(gi.m, 0.3, np.float32), np.zeros(gi.m, np.float32)
    y = np.array(y0, dtype=np.float32)
    solvers.SpmvSolver(solvers.Graph(csr=csr_from(d), in_csr=gi), Ax, x, y)
    np.testing.assert_allclose(y, d["y"], rtol=REL_TOL, atol=1e-30)
    assert orc.spmv_max_rel_error(y, d["y"]) <= 5 * np.sqrt(np.finfo(np.float32).eps)  # SpmvVerifier


@pytest.mark.parametrize("scale,ef,seed", [(15, 16, 11), (18, 16, 12)])
def test_spmv_vs_oracle_rmat(orc, scale, ef, seed):
    g = graphio.rmat_graph(scale, ef, seed=seed)
    gi = graphio.transpose(g)
    rng = np.random.default_rng(seed)
    Ax, x = rng.random(gi.nnz, dtype=np.float32), rng.random(gi.m, dtype=np.float32)
    y0 = rng.random(gi.m, dtype=np.float32)
    want = orc.spmv(gi, Ax, x, y0)
    y = y0.copy()
    solvers.SpmvSolver(solvers.Graph(csr=g, in_csr=gi), Ax, x, y)
    np.testing.assert_allclose(y, want, rtol=REL_TOL, atol=0)
    # linearity: A(2x) == 2 A x exactly in fp32 (power-of-two scaling)
    y2 = np.zeros(gi.m, np.float32)
    y1 = np.zeros(gi.m, np.float32)
    solvers.SpmvSolver(solvers.Graph(csr=g, in_csr=gi), Ax, x, y1)
    solvers.SpmvSolver(solvers.Graph(csr=g, in_csr=gi), Ax, x * np.float32(2), y2)
    assert np.array_equal(y2, y1 * np.float32(2))


def test_spmv_oneshot_on_the_blocked_layout_behind_an_option(orc, monkeypatch):
    """GDN_SPMV_ONESHOT=solve: the one call builds the blocked layout (reported in prep_ms, like the reference's segmenting()
    in front of its Timer, src/spmv/partition.cu:206,269-291) and multiplies on it; same result within the verifier's bound,
    and the default call stays the merge-path pass without a build."""
    g = graphio.rmat_graph(19, 16, seed=21)
    gi = graphio.transpose(g)
    assert gi.nnz >= 1 << 22
    rng = np.random.default_rng(21)
    Ax, x, y0 = rng.random(gi.nnz, dtype=np.float32), rng.random(gi.m, dtype=np.float32), rng.random(gi.m, dtype=np.float32)
    want = orc.spmv(gi, Ax, x, y0)
    G = solvers.Graph(csr=g, in_csr=gi)
    y = y0.copy()
    st0 = solvers.SpmvSolver(G, Ax, x, y)
    np.testing.assert_allclose(y, want, rtol=REL_TOL, atol=0)
    monkeypatch.setenv("GDN_SPMV_ONESHOT", "solve")
    y2 = y0.copy()
    st = solvers.SpmvSolver(G, Ax, x, y2)
    np.testing.assert_allclose(y2, want, rtol=REL_TOL, atol=0)
    assert orc.spmv_max_rel_error(y2, want) <= 5 * np.sqrt(np.finfo(np.float32).eps)
    assert st["prep_ms"] > st0["prep_ms"] and st["edges_traversed"] == gi.nnz


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("scale,ef,seed,signed", [(13, 16, 14, False), (17, 16, 15, True), (15, 64, 16, True)])
def test_spmv_resident_layouts(orc, layout, scale, ef, seed, signed):
    """gdn_spmv_plan_*: CSR merge-path and the propagation-blocked layout (signed fixed-point sums)."""
    g = graphio.rmat_graph(scale, ef, seed=seed)
    gi = graphio.transpose(g)
    rng = np.random.default_rng(seed)
    Ax, x = rng.random(gi.nnz, dtype=np.float32), rng.random(gi.m, dtype=np.float32)
    if signed:
        Ax, x = Ax * np.float32(200) - np.float32(100), x - np.float32(0.5)
    y0 = rng.random(gi.m, dtype=np.float32)
    sp = solvers.ResidentSpMV(solvers.Graph(csr=g, in_csr=gi), Ax, layout)
    for xx in (x, x * np.float32(1e-3), np.zeros_like(x)):
        want = orc.spmv(gi, Ax, xx, y0)
        got = sp.multiply(xx, y0)
        # sums of signed terms cancel: compare against the scale of the row, like SpmvVerifier
        # (maximum_relative_error, src/spmv/spmv_util.h:16-29) but at the north-star 1e-4
        absrow = orc.spmv(gi, np.abs(Ax), np.abs(xx), np.abs(y0))
        assert np.all(np.abs(got - want) <= REL_TOL * absrow + 1e-30)
        assert orc.spmv_max_rel_error(got, want) <= 5 * np.sqrt(np.finfo(np.float32).eps) or signed
    sp.close()


@pytest.mark.parametrize("tiers", [False, True])
def test_spmv_pb_wide_dynamic_range(orc, monkeypatch, tiers):
    """The propagation-blocked layout accumulates in fixed point with ONE scale per multiply.  Rows whose products sit
    far below the largest ones (here: row scales from 1e-30 to 1e+5, x from 1e-12 to 1) lose bits in the conversion --
    phase B hands exactly those rows back and they are recomputed in fp32 like the reference's loop, so every row still
    meets the 1e-4 bound (and SpmvVerifier's criterion); tiny rows must not flush to zero."""
    if tiers:
        monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1")
    g = graphio.rmat_graph(15, 16, seed=71)
    gi = graphio.transpose(g)
    rng = np.random.default_rng(71)
    row_scale = np.float32(10.0) ** rng.integers(-30, 6, gi.m).astype(np.float32)
    rows = np.repeat(np.arange(gi.m), np.diff(gi.rowptr.astype(np.int64)))
    Ax = (rng.random(gi.nnz, dtype=np.float32) + np.float32(0.5)) * row_scale[rows]
    x = (rng.random(gi.m, dtype=np.float32) + np.float32(0.5)) * np.float32(10.0) ** rng.integers(-12, 1, gi.m).astype(np.float32)
    y0 = np.zeros(gi.m, np.float32)
    want = orc.spmv(gi, Ax, x, y0)
    assert (want != 0).sum() > gi.m // 4
    sp = solvers.ResidentSpMV(solvers.Graph(csr=g, in_csr=gi), Ax, layout=1)  # GDN_LAYOUT_PB
    got = sp.multiply(x, y0)
    sp.close()
    nz = want != 0
    assert np.array_equal(got[~nz], want[~nz])
    rel = np.abs(got[nz] - want[nz]) / np.abs(want[nz])
    assert float(rel.max()) < 1e-4, (float(rel.max()), int(nz.nonzero()[0][rel.argmax()]))
    assert orc.spmv_max_rel_error(got, want) <= 5 * np.sqrt(np.finfo(np.float32).eps)


@pytest.mark.parametrize("layout,tiers", [(0, False), (1, False), (1, True)])
def test_spmv_nonfinite_values_propagate_like_the_reference(orc, monkeypatch, layout, tiers):
    """NaN / inf in x (or a product that overflows) must come out of SpmvSolver the way the reference's fp32 loop
    (src/spmv/omp_base.cc:22-33) produces them: NaN rows NaN, inf rows inf, every other row within 1e-4.  The
    merge-path layout (one-shot gdn_spmv, CSR plans) stores the NaN sum; the fixed-point layout cannot hold such a
    product, marks the row and recomputes it in fp32 (it used to take NaN as its internal repair signal and
    dereference a null repair list on the CSR layout)."""
    if tiers:
        monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1")
    g = graphio.rmat_graph(14, 16, seed=73)
    gi = graphio.transpose(g)
    rng = np.random.default_rng(73)
    Ax = rng.random(gi.nnz, dtype=np.float32) + np.float32(0.5)
    x = rng.random(gi.m, dtype=np.float32) + np.float32(0.5)
    deg = np.diff(g.rowptr.astype(np.int64))  # columns of gi = sources of g: pick columns that occur
    used = np.flatnonzero(deg > 0)
    x[used[3]] = np.nan
    x[used[40]] = np.inf
    x[used[-5]] = np.float32(3e38)  # finite, but 3e38 * 1.4 overflows in some rows
    hub = int(np.argmax(deg))
    x[hub] = np.inf  # a column the tiers serve from their table
    y0 = rng.random(gi.m, dtype=np.float32)
    with np.errstate(all="ignore"):
        want = orc.spmv(gi, Ax, x, y0.copy())
    assert np.isnan(want).any() and np.isinf(want).any() and np.isfinite(want).sum() > 0
    if layout == 0:
        got = y0.copy()
        solvers.SpmvSolver(solvers.Graph(csr=g, in_csr=gi), Ax, x, got)
        got2 = None
    sp = solvers.ResidentSpMV(solvers.Graph(csr=g, in_csr=gi), Ax, layout=layout)
    got_plan = sp.multiply(x, y0.copy())
    sp.close()
    for res in ([got, got_plan] if layout == 0 else [got_plan]):
        assert np.array_equal(np.isnan(res), np.isnan(want))
        inf = np.isinf(want)
        assert np.array_equal(res[inf], want[inf])
        fin = np.isfinite(want)
        assert np.isfinite(res[fin]).all()
        rel = np.abs(res[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1e-30)
        assert float(rel.max()) < 1e-4


# ------------------------------------------------------------------ PageRank: rows with very many in-edges
@pytest.mark.parametrize("layout", ["csr", "pb"])
def test_pr_hub_row_deviation_is_bounded_by_the_in_degree(orc, monkeypatch, layout):
    """The reference adds a row's contributions ONE BY ONE in fp32 (src/pr/omp_base.cc:27-33): with n in-edges its sum
    carries up to (n - 1) * 2^-24 relative error (all terms are positive), typically ~sqrt(n) * 2^-24.  This library sums a
    row exactly (2^-62 fixed point, gdn_pb.hpp) or pairwise (merge-path tiles) and rounds once, so on rows with >= 10^4
    in-edges it can differ from `omp_base` by more than the north star's 1e-4 -- towards the exact value.  Pinned here on
    star-heavy rows of 10^3 .. 10^6 in-edges, one iteration from a non-uniform score vector:
      * the GPU score is the fp64 evaluation of the row rounded to fp32 (<= 1 ulp off),
      * it is at least as close to that evaluation as the sequential-fp32 oracle,
      * |GPU - oracle| / oracle <= (n + 2) * 2^-24, the worst case of the reference's own summation order.
    INTEGRATION.md section 2 tells a maintainer; tests/test_gpu_configs.py asserts the same on the hub rows of RMAT-27."""
    monkeypatch.setenv("GDN_PR_LAYOUT", layout)
    monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1")  # record tiers on: hub SOURCES exist too (the rows below feed each other)
    rng = np.random.default_rng(77)
    m = 1 << 21
    hubs = {10 ** 3: 5, 10 ** 4: 6, 10 ** 5: 7, 10 ** 6: 8}  # in-degree -> hub row id
    src_l, dst_l = [], []
    for n, h in hubs.items():
        srcs = rng.choice(np.arange(16, m, dtype=np.int64), size=n, replace=False)
        src_l.append(srcs)
        dst_l.append(np.full(n, h, np.int64))
    # background: every vertex gets a couple of out-edges, so that contributions differ by out-degree
    bg = 4 * m
    src_l.append(rng.integers(0, m, bg))
    dst_l.append(rng.integers(16, m, bg))
    g = graphio.build_csr(m, np.concatenate(src_l), np.concatenate(dst_l))
    gi = graphio.transpose(g)
    deg = g.degrees()
    scores0 = (rng.random(m, dtype=np.float32) + np.float32(0.25)) / np.float32(m)  # non-uniform, sums to ~0.75
    want = scores0.copy()
    orc.pr_iterate(gi, deg, want, 1)  # the reference's loop, one iteration
    got = scores0.copy()
    st = solvers.PRSolver(solvers.Graph(csr=g, in_csr=gi), got, epsilon=0.0, max_iter=1)
    assert st["iterations"] == 2  # MAX_ITER + 1: did not converge (src/pr/omp_base.cc:39 prints iter + 1)
    contrib = np.zeros(m, np.float32)
    nzd = deg > 0
    contrib[nzd] = scores0[nzd] / deg[nzd].astype(np.float32)
    d = np.float32(0.85)
    base = (np.float32(1.0) - d) / np.float32(m)  # src/pr/omp_base.cc:10 base_score, in fp32
    table = {}
    for n, h in hubs.items():
        lo, hi = int(gi.rowptr[h]), int(gi.rowptr[h + 1])
        assert hi - lo >= n
        s64 = float(contrib[gi.colidx[lo:hi]].astype(np.float64).sum())
        exact = np.float32(base + np.float32(d * np.float32(s64)))  # the reference's formula on the exactly summed row
        ulp = float(np.spacing(exact))
        # the fixed-point layout rounds the exact sum once; the merge-path layout adds tile partials pairwise in fp32
        slack = ulp if layout == "pb" else 16 * ulp
        assert abs(float(got[h]) - float(exact)) <= slack, (n, got[h], exact)
        e64 = float(base) + float(d) * s64
        assert abs(float(got[h]) - e64) <= abs(float(want[h]) - e64) + slack, (n, got[h], want[h], e64)
        dev = abs(float(got[h]) - float(want[h])) / float(want[h])
        assert dev <= (hi - lo + 2) * 2.0 ** -24, (n, dev)
        table[n] = dev
    print("hub-row deviation from the sequential-fp32 oracle by in-degree:", {k: "%.2e" % v for k, v in table.items()})
    # every ordinary row (in-degree of a handful) is within the north star's 1e-4
    small = np.diff(gi.rowptr.astype(np.int64)) < 1000
    rel = np.abs(got[small] - want[small]) / want[small]
    assert float(rel.max()) < 1e-4


# ------------------------------------------------------------------ SSSP
@pytest.mark.parametrize("case", ["test_bc_unit", "chesapeake_unit", "rmat10_unit", "rmat10_w255"])
@pytest.mark.parametrize("delta", [1, 7, 1 << 20])
def test_sssp_golden(case, delta):
    d = golden("sssp_" + case)
    g = solvers.Graph(csr=csr_from(d))
    dist = np.full(g.V(), solvers.K_DIST_INF, np.int32)
    solvers.SSSPSolver(g, int(d["source"]), d["weight"], dist, delta)
    assert np.array_equal(dist, d["dist"])


@pytest.mark.parametrize("unit", [False, True])
def test_sssp_oneshot_builds_the_dense_plan_inside_the_call(orc, monkeypatch, unit):
    """SSSPSolver (one call, src/sssp/main.cc:27) builds the blocked layout itself from 2^24 edges on and reports it as
    prep_ms; the threshold is forced down here so that the path runs at test size -- distances exact either way."""
    g = graphio.rmat_graph(16, 16, seed=29)
    rng = np.random.default_rng(29)
    wt = np.ones(g.nnz, np.int32) if unit else rng.integers(1, 256, size=g.nnz).astype(np.int32)
    s = graphio.first_nonisolated(g)
    want = orc.sssp_dijkstra(g, wt, s)
    for dense_min in ("0", "1"):
        monkeypatch.setenv("GDN_SSSP_ONESHOT_DENSE_MIN", dense_min)
        monkeypatch.setenv("GDN_SSSP_DENSE_IN", "100000")  # sweeps from m / 100000 improved rows on: they do run
        dist = np.full(g.m, solvers.K_DIST_INF, np.int32)
        st = solvers.SSSPSolver(solvers.Graph(csr=g), s, wt, dist, 16)
        assert np.array_equal(dist, want), dense_min
        if dense_min == "1":
            assert st["prep_ms"] > 0.0


@pytest.mark.parametrize("scale,ef,seed,delta", [(14, 16, 21, 1), (16, 16, 22, 32), (17, 8, 23, 100)])
def test_sssp_vs_oracle_rmat(orc, scale, ef, seed, delta):
    g = graphio.rmat_graph(scale, ef, seed=seed)
    rng = np.random.default_rng(seed)
    wt = rng.integers(1, 256, size=g.nnz).astype(np.int32)  # GAP convention, generator.h:136
    s = graphio.first_nonisolated(g)
    want = orc.sssp_dijkstra(g, wt, s)
    dist = np.full(g.m, solvers.K_DIST_INF, np.int32)
    solvers.SSSPSolver(solvers.Graph(csr=g), s, wt, dist, delta)
    assert np.array_equal(dist, want)


@pytest.mark.parametrize("scale,ef,seed,wmax", [(12, 16, 24, 1), (16, 16, 25, 255), (18, 16, 26, 1), (17, 32, 27, 1000)])
def test_sssp_resident_dense_sweeps(orc, scale, ef, seed, wmax):
    """gdn_sssp_plan_*: heavy frontiers are relaxed by Bellman-Ford sweeps on the PB layout."""
    g = graphio.rmat_graph(scale, ef, seed=seed)
    rng = np.random.default_rng(seed)
    wt = rng.integers(1, wmax + 1, size=g.nnz).astype(np.int32)
    sp = solvers.ResidentSSSP(solvers.Graph(csr=g), wt, dense=True)
    deg = g.degrees()
    for s, delta in ((graphio.first_nonisolated(g), 1), (int(np.argmax(deg)), max(1, wmax // 4))):
        want = orc.sssp_dijkstra(g, wt, s)
        dist, st = sp.run(s, delta)
        assert np.array_equal(dist, want), (s, int((dist != want).sum()))
    sp.close()


@pytest.mark.parametrize("knobs", [{"GDN_SSSP_BINS": "1"}, {"GDN_SSSP_BINS": "1", "GDN_SSSP_BIN_OUT": "100000000"},
                                   {"GDN_SSSP_BINS": "1", "GDN_SSSP_BIN_CAP": "64"},
                                   {"GDN_SSSP_BINS": "1", "GDN_SSSP_DENSE_IN": "100000", "GDN_SSSP_BIN_OUT": "100000000"}])
@pytest.mark.parametrize("wmax", [1, 255])
def test_sssp_binned_relax_passes(orc, monkeypatch, knobs, wmax):
    """GDN_SSSP_BINS=1 (off by default: measured slower): between the dense sweeps and the worklist tail the plan
    relaxes the out-edges of the rows the last step improved through per-bin lists (sssp_bin_*: propagation blocking made
    on the fly).  Exact distances with the passes taken as long as anything improves (BIN_OUT huge), with lists far too
    short (every pass overflows and is repeated as a sweep), and entered from the first heavy phase on."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    g = graphio.rmat_graph(18, 16, seed=83)
    rng = np.random.default_rng(83)
    wt = rng.integers(1, wmax + 1, size=g.nnz).astype(np.int32)
    sp = solvers.ResidentSSSP(solvers.Graph(csr=g), wt, dense=True)
    for s, delta in ((graphio.first_nonisolated(g), 16), (int(np.argmax(g.degrees())), 1)):
        want = orc.sssp_dijkstra(g, wt, s)
        dist, st = sp.run(s, delta)
        assert np.array_equal(dist, want), (s, int((dist != want).sum()))
        assert st["last_error"] >= st["edges_traversed"]  # edges relaxed >= edges of the reached rows
    sp.close()


@pytest.mark.parametrize("scale,ef,floor,tiers", [(15, 16, "4", None), (17, 8, "2", None), (17, 8, "2", "1"), (16, 32, "64", "3")])
@pytest.mark.parametrize("wlo,whi", [(1, 255), (3, 3), (0, 2)])
def test_sssp_record_tiers_vs_oracle(orc, monkeypatch, capfd, scale, ef, floor, tiers, wlo, whi):
    """The record tiers of the dense sweeps (sssp_build_tiers: the out-edges of the sources of highest out-degree leave the
    blocked layout and are read by phase B as (source index, row) records + 8-bit weights, the source's distance from a
    per-sweep table; from 2^22 edges on) forced onto small graphs: floors of 2 / 4 / 64 out-edges, one tier, and -- scale 17
    with a floor of 2 -- more sources than the first tier holds; weights in 8 bits and all equal (no weight stream), zero
    weights included.  Exact distances for hub, leaf and first sources and two deltas; the sweeps are entered early."""
    monkeypatch.setenv("GDN_SSSP_TIER_MIN_NNZ", "1")
    monkeypatch.setenv("GDN_SSSP_TIER_MIN_DEG", floor)
    monkeypatch.setenv("GDN_SSSP_DENSE_IN", "100000")
    monkeypatch.setenv("GDN_SSSP_TRACE", "1")
    # record streams in lane-interleaved blocks of 256 (the default) -- and plain, as before, in the scale-16 case
    monkeypatch.setenv("GDN_SSSP_REC_IL", "0" if scale == 16 else "1")
    if tiers:
        monkeypatch.setenv("GDN_SSSP_TIERS", tiers)
    g = graphio.rmat_graph(scale, ef, seed=91)
    rng = np.random.default_rng(91)
    wt = rng.integers(wlo, whi + 1, size=g.nnz).astype(np.int32)
    capfd.readouterr()
    sp = solvers.ResidentSSSP(solvers.Graph(csr=g), wt, dense=True)
    log = capfd.readouterr().err
    assert "record tiers" in log, log[-300:]
    if scale == 17 and not tiers:
        assert "[sssp] plan: 2 record tiers" in log or "[sssp] plan: 3 record tiers" in log, log[-300:]
    deg = g.degrees()
    for s, delta in ((graphio.first_nonisolated(g), 16), (int(np.argmax(deg)), 1), (int(np.nonzero(deg == 1)[0][0]), 64)):
        want = orc.sssp_dijkstra(g, wt, s)
        dist, st = sp.run(s, delta)
        assert np.array_equal(dist, want), (s, int((dist != want).sum()))
    sp.close()


def test_sssp_sweeps_that_change_their_candidate_width(orc, monkeypatch):
    """The dense sweeps write 8-bit candidates while every distance + weight stays below 255 and 16-bit ones from then on,
    into one buffer whose alignment gaps are never written and must read as "no path".  A sparse graph whose search passes
    255 under sweeps entered early (graph 6000914 of the randomised sweep, tests/aids/fuzz_parity.py): the rows 0 of some
    bins once took the bytes of old 8-bit candidates for 16-bit ones -- unreachable vertices came back with distances.
    Twice on one plan: the second run starts narrow again on a buffer the first left wide."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "aids", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(6000914)
    g = fz.random_graph(rng)
    source = int(rng.integers(0, g.m))
    wmax = int(rng.choice([2, 16, 256]))
    wt = rng.integers(1, wmax, g.nnz).astype(np.int32)
    monkeypatch.setenv("GDN_SSSP_DENSE_IN", "100000")
    want = orc.sssp_dijkstra(g, wt, source)
    assert want[want != 2147483647].max() > 255 and (want == 2147483647).any()
    sp = solvers.ResidentSSSP(solvers.Graph(csr=g), wt, dense=True)
    for _ in range(2):
        dist, _ = sp.run(source, 1)
        assert np.array_equal(dist, want), int((dist != want).sum())
    sp.close()


def test_old_builder_tiers_under_the_allocation_fence():
    """GDN_ALLOC_FENCE=1 puts every device buffer at the end of its own block of whole 2 MB pages, so a kernel that leaves
    a buffer faults instead of touching a neighbour.  The old layout builder with SSSP's record tiers forced on small
    graphs once scanned one element too many (sssp_build_tiers wrote 8 bytes behind its offsets: a memory fault in about
    one of two 600-graph sweeps, profiles/sessions/r04_82.sh; under the fence graph 26000002 faults at once).  The
    option is read once per process: three graphs of that sweep in a child process."""
    import subprocess
    import sys
    env = dict(os.environ, FUZZ_PLANS="1", GDN_ALLOC_FENCE="1", GDN_PB_BUILDER="old", GDN_PR_LAYOUT="p", GDN_SPMV_LAYOUT="p",
               GDN_PRD_LAYOUT="p", GDN_PB_HUB_MIN_NNZ="1", GDN_PB_HUB_MIN="8", GDN_PB_MID_CAP="300", GDN_BFS_HEADS_MIN_NNZ="1",
               GDN_BFS_HUB_MIN="0", GDN_SSSP_TIER_MIN_NNZ="1", GDN_SSSP_TIER_MIN_DEG="2")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "aids", "fuzz_parity.py"), "3", "26000001"], env=env, cwd=os.path.dirname(here),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "every solver equal to the oracle" in r.stdout, (r.returncode, r.stdout[-400:], r.stderr[-800:])


def test_sssp_wide_weights_keep_the_blocked_layout(orc, monkeypatch, capfd):
    """Weights beyond 8 bits do not fit a record: the plan builds no tiers and every edge stays in the blocked layout."""
    monkeypatch.setenv("GDN_SSSP_TIER_MIN_NNZ", "1")
    monkeypatch.setenv("GDN_SSSP_TIER_MIN_DEG", "2")
    monkeypatch.setenv("GDN_SSSP_DENSE_IN", "100000")
    monkeypatch.setenv("GDN_SSSP_TRACE", "1")
    g = graphio.rmat_graph(14, 16, seed=93)
    wt = np.random.default_rng(93).integers(1, 1000, size=g.nnz).astype(np.int32)
    capfd.readouterr()
    sp = solvers.ResidentSSSP(solvers.Graph(csr=g), wt, dense=True)
    assert "record tiers" not in capfd.readouterr().err
    s = graphio.first_nonisolated(g)
    dist, _ = sp.run(s, 64)
    assert np.array_equal(dist, orc.sssp_dijkstra(g, wt, s))
    sp.close()


@pytest.mark.parametrize("wlo,whi,delta", [(7, 7, 3), (1, 255, 16), (200, 60000, 5000), (1, 1 << 20, 1 << 18), (0, 3, 1),
                                           (1, 3000, 64)])
@pytest.mark.parametrize("knobs", [{}, {"GDN_SSSP_SMALL": "0"}, {"GDN_SSSP_SMALL": "2"}, {"GDN_SSSP_CAND32": "1"},
                                   {"GDN_SSSP_CAND16": "1"}, {"GDN_SSSP_WBYTES": "4"}, {"GDN_SSSP_PAD": "128"},
                                   {"GDN_SSSP_COOP": "1"}, {"GDN_SSSP_COOP": "1", "GDN_SSSP_SMALL": "0"}, {"GDN_SSSP_COOP": "0"}, {"GDN_SSSP_DENSE_IN": "100000", "GDN_SSSP_DENSE_OUT": "100000"}])
def test_sssp_plan_stream_widths_and_fused_phases(orc, monkeypatch, wlo, whi, delta, knobs):
    """The dense sweeps pick the width of their two per-edge streams from the data -- no weight stream at all when every
    weight is equal, 8 / 16 / 32-bit weights, 16-bit candidates while the largest distance + weight stays below 0xFFFF and
    32-bit ones beyond (the (1, 3000) case crosses over in the middle of a solve) -- and the light phases run inside one
    workgroup (GDN_SSSP_SMALL = 0: never, 2: whenever the lists fit at all).  Every combination: exact distances."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    g = graphio.rmat_graph(15, 16, seed=61)
    rng = np.random.default_rng(wlo + whi)
    wt = rng.integers(wlo, whi + 1, size=g.nnz).astype(np.int32)
    sp = solvers.ResidentSSSP(solvers.Graph(csr=g), wt, dense=True)
    deg = g.degrees()
    for s in (graphio.first_nonisolated(g), int(np.argmax(deg))):
        want = orc.sssp_dijkstra(g, wt, s)
        dist, st = sp.run(s, delta)
        assert np.array_equal(dist, want), (s, int((dist != want).sum()))
    sp.close()
    # the plan-less drop-in takes the same fused light phases
    dist = np.full(g.m, solvers.K_DIST_INF, np.int32)
    s = graphio.first_nonisolated(g)
    solvers.SSSPSolver(solvers.Graph(csr=g), s, wt, dist, delta)
    assert np.array_equal(dist, orc.sssp_dijkstra(g, wt, s))


@pytest.mark.parametrize("w", [1, 7, 1 << 27])
def test_sssp_equal_weights_solve_through_the_bfs_plan(orc, monkeypatch, w):
    """Equal weights (the reference main's own input: src/sssp/main.cc:26 fills 1): a resident plan of 2^22 edges or more solves
    through the direction-optimising BFS plan on the transpose it builds and multiplies the depths by the weight
    (GDN_SSSP_UNIT_BFS=1 forces the route at test size, =0 keeps the sweeps).  Same distances as Dijkstra and as the sweeps,
    for any delta, unreached = kDistInf; a product beyond the int range reads "no path" like in the relax kernels."""
    rng = np.random.default_rng(77)
    m, src, dst = graphio.grid2d_edges(60, 60)
    graphs = [graphio.rmat_graph(15, 16, seed=62), graphio.build_csr(m + 5, src, dst)]  # (+ 5 unreachable vertices)
    for g in graphs:
        wt = np.full(g.nnz, w, np.int32)
        res = {}
        for route in ("1", "0"):
            monkeypatch.setenv("GDN_SSSP_UNIT_BFS", route)
            sp = solvers.ResidentSSSP(solvers.Graph(csr=g), wt, dense=True)
            res[route] = [sp.run(s, delta) for s, delta in ((graphio.first_nonisolated(g), 1), (int(np.argmax(g.degrees())), 16))]
            sp.close()
        for (d1, st1), (d0, st0), s in zip(res["1"], res["0"], (graphio.first_nonisolated(g), int(np.argmax(g.degrees())))):
            depth = orc.bfs_serial(g, s).astype(np.int64)
            in_range = bool(((depth == solvers.MYINFINITY) | (depth * w < solvers.K_DIST_INF)).all())
            if in_range:  # (beyond the int range the reference's own `dist + w` is undefined: only the route's contract is checked)
                assert np.array_equal(d1, d0)
                assert st1["edges_traversed"] == st0["edges_traversed"]
            if w < (1 << 20):
                assert np.array_equal(d1, orc.sssp_dijkstra(g, wt, s))
            # depth x 2^27 leaves the int range from depth 16 on: those vertices read kDistInf
            want = np.where((depth == solvers.MYINFINITY) | (depth * w >= solvers.K_DIST_INF), solvers.K_DIST_INF, depth * w)
            assert np.array_equal(d1, want.astype(np.int32))
    # mixed weights never take the route
    monkeypatch.setenv("GDN_SSSP_UNIT_BFS", "1")
    g = graphs[0]
    wt = rng.integers(1, 3, size=g.nnz).astype(np.int32)
    sp = solvers.ResidentSSSP(solvers.Graph(csr=g), wt, dense=True)
    s = graphio.first_nonisolated(g)
    dist, st = sp.run(s, 2)
    sp.close()
    assert np.array_equal(dist, orc.sssp_dijkstra(g, wt, s))


@pytest.mark.parametrize("coop", [None, "1"])
def test_sssp_lattice_runs_on_the_cooperative_grid(orc, monkeypatch, coop):
    """A 200 x 200 lattice (+ a few shortcuts) with U[1,255] weights and small deltas: thousands of buckets of a few
    hundred vertices.  Lists that outgrow the one-workgroup kernel run on the cooperative grid (sssp_coop_kernel: a grid
    barrier per pass and per bucket change); distances exact, from the plan (dense sweeps available) and the drop-in."""
    if coop is not None:
        monkeypatch.setenv("GDN_SSSP_COOP", coop)
    rng = np.random.default_rng(12)
    m, src, dst = graphio.grid2d_edges(200, 200)
    extra = rng.integers(0, m, (50, 2))
    g = graphio.build_csr(m, np.concatenate([src, extra[:, 0]]), np.concatenate([dst, extra[:, 1]]))
    wt = rng.integers(1, 256, size=g.nnz).astype(np.int32)
    sp = solvers.ResidentSSSP(solvers.Graph(csr=g), wt, dense=True)
    for s, delta in ((0, 16), (m // 2 + 13, 64), (m - 1, 3)):
        want = orc.sssp_dijkstra(g, wt, s)
        dist, st = sp.run(s, delta)
        assert np.array_equal(dist, want), (s, delta, int((dist != want).sum()))
        d2 = np.full(m, solvers.K_DIST_INF, np.int32)
        solvers.SSSPSolver(solvers.Graph(csr=g), s, wt, d2, delta)
        assert np.array_equal(d2, want)
    sp.close()


def test_sssp_long_weighted_path_runs_inside_one_workgroup(orc):
    """A 30 000-vertex path with a few shortcuts and delta = 1: tens of thousands of buckets of one or two vertices.  Every
    pass and bucket change stays on the device (stats.iterations counts the passes)."""
    n = 30000
    src = np.concatenate([np.arange(n - 1), np.arange(0, n - 50, 50)])
    dst = np.concatenate([np.arange(1, n), np.arange(40, n - 10, 50)])
    g = graphio.build_csr(n, src, dst)
    rng = np.random.default_rng(3)
    wt = rng.integers(1, 4, size=g.nnz).astype(np.int32)
    want = orc.sssp_dijkstra(g, wt, 0)
    for delta in (1, 5):
        dist = np.full(n, solvers.K_DIST_INF, np.int32)
        st = solvers.SSSPSolver(solvers.Graph(csr=g), 0, wt, dist, delta)
        assert np.array_equal(dist, want)
        assert st["iterations"] > 1000 and st["solve_ms"] < 2000


# ------------------------------------------------------------------ CC
@pytest.mark.parametrize("case", ["test_cc_sym", "chesapeake_sym", "rmat10_sym", "rmat10_dir"])
def test_cc_golden(orc, case):
    d = golden("cc_" + case)
    sym = bool(d["symmetrize"])
    g = solvers.Graph(csr=csr_from(d), in_csr=csr_from(d, "in_"), symmetrize=sym, need_reverse=not sym)
    comp = np.arange(g.V(), dtype=np.int32)
    solvers.CCSolver(g, comp)
    assert np.array_equal(comp, d["comp_sv"])
    if sym:
        assert orc.cc_verify(csr_from(d), comp)  # CCVerifier criterion


# (scale 20: more than 2^18 vertices -- the sampling rounds after the first then run in two launches, gdn_cc.hip)
@pytest.mark.parametrize("scale,ef,seed", [(14, 4, 31), (16, 16, 32), (18, 2, 33), (20, 6, 34)])
@pytest.mark.parametrize("variant", ["afforest_sym", "afforest_directed", "sv_no_reverse"])
def test_cc_vs_oracle_rmat(orc, scale, ef, seed, variant):
    d = graphio.rmat_graph(scale, ef, seed=seed)
    gs = graphio.symmetrize(d)
    want, _ = orc.cc_sv(gs)  # weakly connected components, min-id labels
    if variant == "afforest_sym":      # symmetric graph, reverse = alias  -> Afforest
        G, g = solvers.Graph(csr=gs, symmetrize=True), gs
    elif variant == "afforest_directed":  # directed graph + reverse graph -> Afforest over out- and in-edges
        G, g = solvers.Graph(csr=d, need_reverse=True), d
    else:                               # directed graph, no reverse     -> symmetric SV hook
        G, g = solvers.Graph(csr=d), d
    comp = np.arange(g.m, dtype=np.int32)
    solvers.CCSolver(G, comp)
    assert np.array_equal(comp, want)
    # min-id property
    assert np.all(comp <= np.arange(g.m)) and np.all(comp[comp] == comp)


@pytest.mark.parametrize("sv", ["1", "fused"])
def test_cc_shiloach_vishkin_rounds_and_their_fused_kernel(orc, monkeypatch, sv):
    """GDN_CC_SV=1: the reference's own algorithm (hook + pointer jumping until nothing changes, src/cc/omp_base.cc:19-44) as
    launches; GDN_CC_SV=fused: the same rounds inside ONE cooperative kernel behind the grid barrier -- the counterpart of
    src/cc/fusion.cu:47 cc_kernel.  Both end at the minimum-id labels of the oracle's SV fixpoint on directed (out-edges only),
    symmetric and many-component graphs, a path of 3 000 vertices (many rounds) included."""
    monkeypatch.setenv("GDN_CC_SV", sv)
    rng = np.random.default_rng(91)
    path = np.arange(2999)
    graphs = [graphio.rmat_graph(15, 16, seed=92), graphio.symmetrize(graphio.rmat_graph(13, 4, seed=93)),
              graphio.build_csr(3000, path, path + 1),
              graphio.build_csr(5000, rng.integers(0, 5000, 3000), rng.integers(0, 5000, 3000))]
    for g in graphs:
        want, _ = orc.cc_sv(graphio.symmetrize(g))
        comp = np.arange(g.m, dtype=np.int32)
        st = solvers.CCSolver(solvers.Graph(csr=g), comp)
        assert np.array_equal(comp, want)
        assert st["iterations"] >= 2 and st["reserved"] == (2 if sv == "fused" else 0)


def test_cc_out_edges_only_with_the_reverse_graph_built_in_the_call(orc, monkeypatch):
    """GDN_CC_REVERSE=build: a directed graph handed over without its reverse gets the transpose built inside gdn_cc_dev
    (charged to prep_ms) and the solve WITH it; same minimum-id labels as the default, stats.reserved = 3."""
    rng = np.random.default_rng(95)
    graphs = [graphio.rmat_graph(16, 16, seed=96), graphio.build_csr(5000, rng.integers(0, 5000, 3000), rng.integers(0, 5000, 3000)),
              graphio.build_csr(40, np.zeros(0, np.int64), np.zeros(0, np.int64))]
    for g in graphs:
        want, _ = orc.cc_sv(graphio.symmetrize(g))
        plain = np.arange(g.m, dtype=np.int32)
        st0 = solvers.CCSolver(solvers.Graph(csr=g), plain)
        monkeypatch.setenv("GDN_CC_REVERSE", "build")
        comp = np.arange(g.m, dtype=np.int32)
        st = solvers.CCSolver(solvers.Graph(csr=g), comp)
        monkeypatch.delenv("GDN_CC_REVERSE")
        assert np.array_equal(comp, want) and np.array_equal(plain, want)
        assert st0["reserved"] == 0 and st["reserved"] == (3 if g.nnz else 0)
        if g.nnz:
            assert st["prep_ms"] > 0


def test_cc_many_small_components(orc):
    """No giant component: Afforest's sampled label covers almost nothing, every edge is walked."""
    n, k = 60000, 6  # 10000 cycles of 6 vertices + a long path
    base = np.arange(0, n, k)
    src = np.concatenate([base + i for i in range(k)])
    dst = np.concatenate([base + (i + 1) % k for i in range(k)])
    path = np.arange(n, n + 3000)
    src = np.concatenate([src, path[:-1]])
    dst = np.concatenate([dst, path[1:]])
    g = graphio.symmetrize(graphio.build_csr(n + 3000, src, dst))
    want, _ = orc.cc_sv(g)
    comp = np.arange(g.m, dtype=np.int32)
    solvers.CCSolver(solvers.Graph(csr=g, symmetrize=True), comp)
    assert np.array_equal(comp, want)


def test_cc_directed_without_reverse_links_every_out_edge(orc):
    """A directed graph handed over WITHOUT its reverse (gdn_cc's in_* = NULL): two sampling rounds, then one pass over the
    remaining out-edges of every vertex.  Stars whose leaves have no out-edge of their own hang on the hub's row alone --
    also its LAST edges, which a big row (cut into work items from its first edge) once lost when the two sampled
    neighbours were skipped."""
    rng = np.random.default_rng(5)
    hubs = [(0, 700), (1, 513), (2, 514), (3, 3000), (4, 63), (5, 2)]
    src, dst, nxt = [], [], 10
    for h, k in hubs:
        src += [h] * k
        dst += list(range(nxt, nxt + k))
        nxt += k
    m = nxt + 1000
    extra = rng.integers(nxt, m, (2000, 2))  # a random sparse part behind the stars
    src += extra[:, 0].tolist()
    dst += extra[:, 1].tolist()
    g = graphio.build_csr(m, np.array(src, np.int64), np.array(dst, np.int64))
    want, _ = orc.cc_sv(g)
    comp = np.arange(m, dtype=np.int32)
    solvers.CCSolver(solvers.Graph(csr=g), comp)  # no reverse graph
    assert np.array_equal(comp, want)
    lo = 10
    for h, k in hubs:
        assert np.all(comp[lo:lo + k] == h)
        lo += k


# ------------------------------------------------------------------ TC
@pytest.mark.parametrize("form", ["f", "a", "u", "v", "bs"])
@pytest.mark.parametrize("case", ["chesapeake_sym", "rmat10_sym"])
def test_tc_golden(case, form, monkeypatch):
    """Every formulation of the count -- the forward count on the rank-ordered DAG (default), the LDS hash set walked u- or
    v-centric on the reference's orientation, and the north star's wave-per-edge binary-search intersect (GDN_TC_FORM=bs:
    src/tc/gpu_base.cu:11-23 re-cut for wave64) -- gives the reference's total."""
    monkeypatch.setenv("GDN_TC_FORM", form)
    d = golden("tc_" + case)
    total, st = solvers.TCSolver(solvers.Graph(csr=csr_from(d, "sym_"), symmetrize=True))
    assert total == int(d["total"])
    assert st["edges_traversed"] == len(d["dag_colidx"])
    total2, _ = solvers.TCSolver(solvers.Graph(csr=csr_from(d, "dag_")), oriented=True)
    assert total2 == int(d["total"])


def test_pr_one_shot_takes_the_blocked_layout_from_2_22_edges_on(orc, monkeypatch):
    """the PRSolver drop-in (gdn_pr, no plan handed in) runs the propagation-blocked layout by default on a graph of
    >= 2^22 edges -- the layout build is ~100 ps per edge since round 4, a solve to 1e-4 saves more than that --; GDN_PR_LAYOUT
    forces with c / p and means `auto` with anything else; small graphs stay on the caller's CSR."""
    g = graphio.rmat_graph(19, 16, seed=5)
    assert g.nnz >= 1 << 22
    G = solvers.Graph(csr=g, need_reverse=True)
    want, it, _ = orc.pr(graphio.transpose(g), np.diff(g.rowptr.astype(np.int64)).astype(np.int32))
    for env, lay in ((None, "pb"), ("c", "csr"), ("p", "pb"), ("auto", "pb")):
        if env is None:
            monkeypatch.delenv("GDN_PR_LAYOUT", raising=False)
        else:
            monkeypatch.setenv("GDN_PR_LAYOUT", env)
        s = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        st = solvers.PRSolver(G, s)
        assert st["layout"] == lay and st["iterations"] == it and np.allclose(s, want, rtol=1e-4, atol=0), (env, st["layout"])
    monkeypatch.delenv("GDN_PR_LAYOUT", raising=False)
    small = graphio.rmat_graph(14, 8, seed=5)
    s = np.full(small.m, np.float32(1.0) / np.float32(small.m), np.float32)
    assert solvers.PRSolver(solvers.Graph(csr=small, need_reverse=True), s)["layout"] == "csr"


@pytest.mark.parametrize("form", ["", "f"])
def test_tc_oriented_input_that_is_no_dag_counts_like_the_reference_loop(orc, form, monkeypatch):
    """with oriented=1 the count runs on the caller's lists AS THEY ARE (src/tc/omp_base.cc:16-22 does not know
    whether they are an orientation): a directed graph with 2-cycles and cyclic triangles gives the reference loop's total,
    also when the forward form is asked for (the default from 2^22 DAG edges on) -- its precondition check (every edge
    ascends in the (degree, id) order, no entry repeats) sends such an input to the count on the given lists."""
    if form:
        monkeypatch.setenv("GDN_TC_FORM", form)
    g = graphio.rmat_graph(12, 12, seed=77)  # directed: both directions of some pairs, cyclic triangles
    want = orc.tc(g)
    src, dst = graphio.csr_to_coo(graphio.symmetrize(g))
    assert want != orc.tc(orc.tc_orient(graphio.symmetrize(g)))  # (not the triangle count of the underlying graph)
    total, st = solvers.TCSolver(solvers.Graph(csr=g), oriented=True)
    assert total == want and st["reserved"] in (0, 1)
    # the reference's own orientation still takes the forward form when asked to
    dag = orc.tc_orient(graphio.symmetrize(g))
    total, st = solvers.TCSolver(solvers.Graph(csr=dag), oriented=True)
    assert total == orc.tc(dag) and st["reserved"] == (3 if form else st["reserved"])


@pytest.mark.parametrize("core", ["4096", "8192", "12288", "16384"])
@pytest.mark.parametrize("scale,ef,seed", [(13, 16, 41), (15, 32, 43), (17, 8, 44)])
def test_tc_forward_core_vs_oracle(orc, scale, ef, seed, core, monkeypatch):
    """The forward count with its CORE: the middle vertices v among the top K ranks are counted on the K x K bit matrix
    (tc_core_count_kernel: a row AND per (u, v), pair tests for the u with few core neighbours), the rows below the core
    by the hash-set kernel -- together the forward count, for every K (the default takes a core of 16384 ranks from
    2^21 vertices on).  RMAT-13 with K = 8192 / 16384: the graph is smaller than the core + 64 -- no core, the plain forward count."""
    monkeypatch.setenv("GDN_TC_FORM", "f")
    monkeypatch.setenv("GDN_TC_CORE", core)
    # (the walks' packed bounds -- first element << 24 | elements, read beside the neighbour ids instead of two gathered row
    # offsets; default from 2^22 DAG edges on -- with two of the four core sizes)
    monkeypatch.setenv("GDN_TC_NBOUND", "1" if core in ("8192", "16384") else "0")
    g = graphio.symmetrize(graphio.rmat_graph(scale, ef, seed=seed))
    want = orc.tc(orc.tc_orient(g))
    total, st = solvers.TCSolver(solvers.Graph(csr=g, symmetrize=True))
    has_core = g.m >= int(core) + 64
    assert total == want and st["reserved"] & 0xFF == 3 and st["reserved"] >> 8 == (int(core) if has_core else 0), (total, want, st)
    dag = orc.tc_orient(g)
    total, st = solvers.TCSolver(solvers.Graph(csr=dag), oriented=True)
    assert total == want and st["reserved"] >> 8 == (int(core) if has_core else 0)


@pytest.mark.parametrize("core", ["0", "4096"])
def test_tc_plan_walked_elements(monkeypatch, core):
    """gdn_tc_plan_walked_elements (bench.py's kernel_list_read_frac): the list elements the forward count walks against its hash
    sets = SUM over the DAG edges u -> v whose middle vertex v ranks below the core of the out-neighbours of u that outrank v;
    without a core (SUM d+(u)^2 - nnz) / 2."""
    import ctypes as C
    from gardenia_amd import _cabi
    monkeypatch.setenv("GDN_TC_FORM", "f")
    monkeypatch.setenv("GDN_TC_CORE", core)
    g = graphio.symmetrize(graphio.rmat_graph(14, 16, seed=45))
    deg = g.degrees().astype(np.int64)
    rank = np.empty(g.m, np.int64)
    rank[np.lexsort((np.arange(g.m), deg))] = np.arange(g.m)  # (degree, id) order: src/common/graph.cc:80-81
    src, dst = graphio.csr_to_coo(g)
    keep = rank[dst] > rank[src]
    ru, rv = rank[src[keep]], rank[dst[keep]]
    order = np.lexsort((rv, ru))
    ru, rv = ru[order], rv[order]
    dplus = np.bincount(ru, minlength=g.m)
    first = np.concatenate(([0], np.cumsum(dplus)[:-1]))
    pos = np.arange(len(ru)) - first[ru]            # position of v in u's ascending list
    base = g.m - int(core)
    want = int((dplus[ru] - 1 - pos)[rv < base].sum())
    if core == "0":
        assert want == (int((dplus ** 2).sum()) - len(ru)) // 2
    L = _cabi.lib()
    h, plan = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_graph_upload(g.m, g.nnz, g.rowptr.ctypes.data_as(C.c_void_p), g.colidx.ctypes.data_as(C.c_void_p), C.byref(h)))
    _cabi.check(L.gdn_tc_plan_create(h, 0, C.byref(plan)))
    got = C.c_uint64(0)
    _cabi.check(L.gdn_tc_plan_walked_elements(plan, C.byref(got)))
    total, st = C.c_uint64(0), _cabi.GdnStats()
    _cabi.check(L.gdn_tc_plan_count(plan, C.byref(total), C.byref(st)))
    L.gdn_tc_plan_free(plan)
    L.gdn_graph_free(h)
    assert st.reserved >> 8 == int(core) and got.value == want > 0


def test_tc_forward_core_dense_and_sparse_shapes(orc, monkeypatch):
    """Shapes the core meets at its edges: a clique larger than the pair-test limit inside a sparse graph (every member's
    core list takes the row path), a star forest (core neighbours but no triangle), a graph whose top ranks are ALL of
    equal degree (ties broken by id), and one with exactly core + 64 vertices."""
    monkeypatch.setenv("GDN_TC_FORM", "f")
    monkeypatch.setenv("GDN_TC_CORE", "4096")
    rng = np.random.default_rng(808)
    m = 4096 + 64
    cl = rng.choice(m, 150, replace=False)
    a, b = np.meshgrid(cl, cl)
    src = [a[a != b].ravel()]
    dst = [b[a != b].ravel()]
    hubs = rng.choice(m, 40, replace=False)  # stars: many leaves around a few centres
    leaves = rng.integers(0, m, 6000)
    hub_of = hubs[rng.integers(0, 40, 6000)]
    keep = leaves != hub_of
    src += [leaves[keep], hub_of[keep]]
    dst += [hub_of[keep], leaves[keep]]
    ring = np.arange(m)  # everybody has degree >= 4: two rings
    for step in (1, 7):
        src += [ring, (ring + step) % m]
        dst += [(ring + step) % m, ring]
    g = graphio.symmetrize(graphio.build_csr(m, np.concatenate(src).astype(np.int64), np.concatenate(dst).astype(np.int64)))
    want = orc.tc(orc.tc_orient(g))
    assert want > 150 * 149 * 148 // 6 - 1
    total, st = solvers.TCSolver(solvers.Graph(csr=g, symmetrize=True))
    assert total == want and st["reserved"] >> 8 == 4096, (total, want)
    for knob, value in (("GDN_TC_CORE_SMALL", "2"), ("GDN_TC_CORE_SMALL", "64"), ("GDN_TC_CORE_ASYNC", "0"), ("GDN_TC_CORE_WGS", "8"), ("GDN_TC_CORE_TAIL", "6"),
                        ("GDN_TC_NBOUND", "1")):
        monkeypatch.setenv(knob, value)  # every core list as rows / the short ones as pairs; in front of the hash-set kernel
        total, st = solvers.TCSolver(solvers.Graph(csr=g, symmetrize=True))
        assert total == want and st["reserved"] >> 8 == 4096, (knob, value, total, want)
        monkeypatch.delenv(knob)
    monkeypatch.setenv("GDN_TC_CORE", "0")
    total, st = solvers.TCSolver(solvers.Graph(csr=g, symmetrize=True))
    assert total == want and st["reserved"] == 3


@pytest.mark.parametrize("scale,ef,seed", [(13, 16, 41), (16, 8, 42)])
def test_tc_vs_oracle_rmat(orc, scale, ef, seed, monkeypatch):
    g = graphio.symmetrize(graphio.rmat_graph(scale, ef, seed=seed))
    want = orc.tc(orc.tc_orient(g))
    total, st = solvers.TCSolver(solvers.Graph(csr=g, symmetrize=True))
    assert total == want and st["reserved"] in (0, 1)  # below 2^22 DAG edges: the hash-set count on the reference's orientation
    monkeypatch.setenv("GDN_TC_FORM", "f")  # the forward count (the default from 2^22 DAG edges on)
    total, st = solvers.TCSolver(solvers.Graph(csr=g, symmetrize=True))
    assert total == want and st["reserved"] == 3
    dag = orc.tc_orient(g)
    total, st = solvers.TCSolver(solvers.Graph(csr=dag), oriented=True)  # handed a DAG: re-ranked from in + out degrees
    assert total == want and st["edges_traversed"] == dag.nnz
    src, dst = graphio.csr_to_coo(g)  # ANY acyclic orientation may be handed over: here "towards the higher id"
    by_id = graphio.build_csr(g.m, src[dst > src], dst[dst > src])
    total, _ = solvers.TCSolver(solvers.Graph(csr=by_id), oriented=True)
    assert total == want
    for form in ("u", "v"):
        monkeypatch.setenv("GDN_TC_FORM", form)
        total, st = solvers.TCSolver(solvers.Graph(csr=g, symmetrize=True))
        assert total == want and st["reserved"] == (1 if form == "v" else 0)
    monkeypatch.setenv("GDN_TC_FORM", "bs")  # the binary-search intersect: lists from 0 to ~10^3 ids, pivots + segments
    total, st = solvers.TCSolver(solvers.Graph(csr=g, symmetrize=True))
    assert total == want and st["reserved"] == 2
    # the plan API: prepared once (the preparation the reference does while loading), counted twice
    import ctypes as C
    from gardenia_amd import _cabi
    L = _cabi.lib()
    for form in ("f", "a"):
        monkeypatch.setenv("GDN_TC_FORM", form)
        h, plan = C.c_void_p(), C.c_void_p()
        rp, ci = np.ascontiguousarray(g.rowptr, np.uint64), np.ascontiguousarray(g.colidx, np.int32)
        _cabi.check(L.gdn_graph_upload(g.m, g.nnz, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), C.byref(h)))
        _cabi.check(L.gdn_tc_plan_create(h, 0, C.byref(plan)))
        L.gdn_graph_free(h)  # the plan keeps nothing of the caller's graph
        for _ in range(2):
            t, ps = C.c_uint64(0), _cabi.GdnStats()
            _cabi.check(L.gdn_tc_plan_count(plan, C.byref(t), C.byref(ps)))
            assert t.value == want and ps.edges_traversed == dag.nnz and ps.solve_ms > 0
        L.gdn_tc_plan_free(plan)


# ------------------------------------------------------------------ device graph builder
def test_device_rmat_matches_numpy_generator():
    import ctypes as C
    from gardenia_amd import _cabi
    L = _cabi.lib()
    for scale, ef in [(10, 16), (14, 8)]:
        want = graphio.rmat_graph(scale, ef)
        want_in = graphio.transpose(want)
        go, gi = C.c_void_p(), C.c_void_p()
        _cabi.check(L.gdn_rmat_build(scale, ef, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
        for h, w in ((go, want), (gi, want_in)):
            m, nnz = C.c_int32(), C.c_uint64()
            _cabi.check(L.gdn_graph_info(h, C.byref(m), C.byref(nnz), None, None))
            assert (m.value, nnz.value) == (w.m, w.nnz)
            rp, ci = np.empty(w.m + 1, np.uint64), np.empty(w.nnz, np.int32)
            _cabi.check(L.gdn_graph_download(h, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p)))
            assert np.array_equal(rp, w.rowptr) and np.array_equal(ci, w.colidx)
        gt = C.c_void_p()
        _cabi.check(L.gdn_graph_transpose(go, C.byref(gt)))
        rp, ci = np.empty(want.m + 1, np.uint64), np.empty(want.nnz, np.int32)
        _cabi.check(L.gdn_graph_download(gt, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(rp, want_in.rowptr) and np.array_equal(ci, want_in.colidx)
        for h in (go, gi, gt):
            L.gdn_graph_free(h)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("layout", [0, 1])
def test_spmv_row_shards_on_one_device(orc, world, layout):
    """gdn_spmv_plan_create_cols: the plan of a vertex-range row shard (global column ids, x of the global
    length) -- the local multiply of gardenia_amd.sharded.ShardedSpMV -- against the oracle's whole product."""
    import ctypes as C
    from gardenia_amd import _cabi
    from gardenia_amd.sharded import vertex_range
    L = _cabi.lib()
    g = graphio.rmat_graph(15, 16, seed=41)
    m = g.m - 3
    src, dst = graphio.csr_to_coo(g)
    keep = (src < m) & (dst < m)
    g = graphio.build_csr(m, src[keep], dst[keep])
    rng = np.random.default_rng(8)
    Ax = (rng.random(g.nnz) - 0.5).astype(np.float32)
    x = (rng.random(m) - 0.5).astype(np.float32)
    y0 = rng.random(m).astype(np.float32)
    want = orc.spmv(g, Ax, x, y0)

    def dev(a):
        p = C.c_void_p()
        _cabi.check(L.gdn_dev_alloc(max(a.nbytes, 4), C.byref(p)))
        if a.nbytes:
            _cabi.check(L.gdn_dev_upload(p, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return p

    h = C.c_void_p()
    _cabi.check(L.gdn_graph_upload(m, g.nnz, g.rowptr.ctypes.data_as(C.c_void_p), g.colidx.ctypes.data_as(C.c_void_p),
                                   C.byref(h)))
    d_x = dev(x)
    got = np.empty(m, np.float32)
    for r in range(world):
        lo, hi, _ = vertex_range(r, world, m)
        sh, plan = C.c_void_p(), C.c_void_p()
        _cabi.check(L.gdn_graph_slice_rows(h, lo, hi, C.byref(sh)))
        e0, e1 = int(g.rowptr[lo]), int(g.rowptr[hi])
        d_Ax, d_y = dev(np.ascontiguousarray(Ax[e0:e1])), dev(np.ascontiguousarray(y0[lo:hi]))
        _cabi.check(L.gdn_spmv_plan_create_cols(sh, d_Ax, m, layout, C.byref(plan)))
        _cabi.check(L.gdn_spmv_dev(plan, d_Ax, d_x, d_y, None))
        _cabi.check(L.gdn_spmv_plan_check(plan))
        part = np.empty(hi - lo, np.float32)
        _cabi.check(L.gdn_dev_download(part.ctypes.data_as(C.c_void_p), d_y, part.nbytes))
        got[lo:hi] = part
        L.gdn_spmv_plan_free(plan)
        L.gdn_graph_free(sh)
        L.gdn_dev_free(d_Ax)
        L.gdn_dev_free(d_y)
    L.gdn_dev_free(d_x)
    L.gdn_graph_free(h)
    assert orc.spmv_max_rel_error(got, want) <= 5 * np.sqrt(np.finfo(np.float32).eps)  # src/spmv/verifier.cc:24
    np.testing.assert_allclose(got, want, rtol=REL_TOL, atol=1e-6)


# ------------------------------------------------------------------ BC
BC_ATOL = BC_RTOL = 1e-4  # the reference verifier's criterion (src/bc/verifier.cc:22-23)


def _bc_close(got, want):
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    g, w = got[~nan].astype(np.float64), want[~nan].astype(np.float64)
    assert np.all(np.abs(g - w) <= BC_RTOL * (np.abs(g) + np.abs(w)) + BC_ATOL)


@pytest.mark.parametrize("case", ["test_bc_dir", "test_bc_sym", "chesapeake_sym", "4_dir", "rmat10_dir", "rmat12_dir"])
def test_bc_golden(orc, case):
    """Scores the reference itself produced (src/bc/omp_base.cc through oracle/_ref/ref_bc): within the reference
    verifier's tolerance everywhere, and bit for bit on the graphs whose rows all have fewer than 32 out-edges (longer
    rows are summed by a wave in a fixed tree, not in CSR order)."""
    d = golden("bc_" + case)
    g = solvers.Graph(csr=csr_from(d))
    scores = np.zeros(g.V(), np.float32)
    st = solvers.BCSolver(g, int(d["source"]), scores)
    _bc_close(scores, d["scores"])
    if int(csr_from(d).degrees().max()) < 32:  # every row summed by one lane in CSR order: the reference's bits
        assert np.array_equal(scores, d["scores"], equal_nan=True)
    assert st["iterations"] == orc.bc(csr_from(d), int(d["source"]))[1]
    assert orc.bc_verify(csr_from(d), int(d["source"]), scores)


@pytest.mark.parametrize("scale,ef,seed,sym", [(14, 16, 1, False), (16, 16, 2, False), (15, 8, 3, True), (18, 16, 4, False)])
def test_bc_vs_oracle_rmat(orc, scale, ef, seed, sym):
    g = graphio.rmat_graph(scale, ef, seed=seed)
    if sym:
        g = graphio.symmetrize(g)
    s = graphio.first_nonisolated(g)
    want, levels, depths, pcs = orc.bc(g, s)
    scores = np.zeros(g.m, np.float32)
    st = solvers.BCSolver(solvers.Graph(csr=g), s, scores)
    assert st["iterations"] == levels
    _bc_close(scores, want)
    assert scores.max() == 1.0 and np.all(scores[depths == -1] == 0.0)
    assert st["edges_traversed"] == 2 * int(g.degrees()[depths >= 0].astype(np.int64).sum())


def test_bc_accumulates_into_scores_and_handles_hubs(orc):
    """scores is in/out (scores[v] += delta[v] before the normalisation, src/bc/omp_base.cc:91); a star with 10 000
    leaves and a second layer exercises the workgroup-per-row path of the backward sweep (rows >= 4096 edges)."""
    n_leaf, n_far = 10000, 300
    src = np.concatenate([np.zeros(n_leaf, np.int64), np.arange(1, n_far + 1), np.arange(1, n_far + 1)])
    dst = np.concatenate([np.arange(1, n_leaf + 1), n_leaf + 1 + np.arange(n_far), n_leaf + 1 + (np.arange(n_far) + 1) % n_far])
    g = graphio.build_csr(n_leaf + 1 + n_far, src, dst)
    start = np.random.default_rng(2).random(g.m).astype(np.float32)
    want, _, _, _ = orc.bc(g, 0, scores=start)
    got = start.copy()
    solvers.BCSolver(solvers.Graph(csr=g), 0, got)
    _bc_close(got, want)
    # an isolated source: nothing is reached, every score is 0/0 like in the reference
    g2 = graphio.build_csr(5, np.array([1, 2]), np.array([2, 3]))
    sc = np.zeros(5, np.float32)
    solvers.BCSolver(solvers.Graph(csr=g2), 0, sc)
    assert np.isnan(sc).all() and np.isnan(orc.bc(g2, 0)[0]).all()


@pytest.mark.parametrize("scale,ef,seed,sym,with_reverse", [(16, 16, 2, False, True), (15, 8, 3, True, False), (18, 16, 4, False, True)])
def test_bc_plan_vs_oracle_rmat(orc, scale, ef, seed, sym, with_reverse):
    """gdn_bc_plan_*: depths from the BFS plan, heavy levels as propagation-blocked sweeps (on R-MAT graphs of this size
    the levels 2..3 are heavy: more than 1/16 of the edges).  Against the oracle within the reference verifier's
    tolerance, several sources on one plan, the same level count, NaN-free where the oracle is."""
    g = graphio.rmat_graph(scale, ef, seed=seed)
    if sym:
        g = graphio.symmetrize(g)
    G = solvers.Graph(csr=g, in_csr=g if sym else graphio.transpose(g))
    bc = solvers.ResidentBC(G, with_reverse=with_reverse)
    deg = g.degrees()
    sources = [graphio.first_nonisolated(g), int(np.argmax(deg)), int(np.flatnonzero(deg > 0)[-1])]
    for s in sources:
        want, levels, depths, pcs = orc.bc(g, s)
        scores = np.zeros(g.m, np.float32)
        st = bc.run(s, scores)
        assert st["iterations"] == levels
        _bc_close(scores, want)
        assert orc.bc_verify(g, s, scores)
        assert st["edges_traversed"] == 2 * int(deg[depths >= 0].astype(np.int64).sum())
    # the queue-based path and the plan agree within the same tolerance
    scores2 = np.zeros(g.m, np.float32)
    solvers.BCSolver(solvers.Graph(csr=g), sources[-1], scores2)
    _bc_close(scores, scores2)
    bc.close()


# ------------------------------------------------------------------ delta PageRank
def _prd_check(scores, tr, st, want, it, wtr):
    """Same pull/push decisions and iteration count as the oracle; scores within 1e-4 relative.  The frontier test
    |delta| > 1e-3 * score sits on fp32 values that differ from the sequential sums by an ulp or two, so a vertex ON the
    threshold may fall on the other side: its delta (1e-3 of its score) is then pushed or not, which is why the scores
    beyond 1e-4 are allowed, IN ALL, the size of four such terms and the frontier sizes a small slack."""
    assert st["iterations"] == it
    assert np.array_equal(tr["mode"], wtr["mode"])
    assert np.all(np.abs(tr["items"].astype(np.int64) - wtr["items"]) <= 2 + wtr["items"] // 2000)
    np.testing.assert_allclose(tr["diff"], wtr["diff"], rtol=1e-4, atol=1e-7)
    rel = np.abs(scores - want) / np.maximum(np.abs(want), 1e-30)
    bad = rel > REL_TOL
    # what up to four flipped vertices can move in all: each pushes (or not) 0.85 * delta, |delta| ~ 1e-3 * its score,
    # spread over its out-neighbours (tests/aids/fuzz_parity.py seed 176: one flip, five neighbours 2e-4 off)
    assert float(np.abs(scores - want)[bad].sum()) <= 4 * 0.85e-3 * float(want.max()), (int(bad.sum()), rel.max())
    assert rel.max() < 2e-3


@pytest.mark.parametrize("case", ["test_pr", "chesapeake_sym", "test_bc_dir", "rmat10", "rmat12"])
def test_pr_delta_golden(case):
    """src/pr/omp_delta.cc run by the reference build itself (tests/golden/make_golden.py pr_delta)."""
    d = golden("prdelta_" + case)
    g = solvers.Graph(csr=csr_from(d), in_csr=csr_from(d, "in_"))
    scores = np.full(g.V(), np.float32(1.0) / np.float32(g.V()), np.float32)
    st = solvers.PRDeltaSolver(g, scores, push_div=10)
    assert st["iterations"] + 1 == int(d["iterations_printed"])  # omp_delta.cc:105 prints iter + 1
    np.testing.assert_allclose(scores, d["scores"], rtol=REL_TOL, atol=0)
    assert abs(st["last_error"] - d["trace"][-1]) < 2e-6


@pytest.mark.parametrize("scale,ef,seed", [(14, 16, 5), (17, 16, 6), (12, 64, 7)])
@pytest.mark.parametrize("push_div", [8, 10, 2])
@pytest.mark.parametrize("heavy_div", ["1", "64", "1000000000"])
def test_pr_delta_vs_oracle_rmat(orc, monkeypatch, scale, ef, seed, push_div, heavy_div):
    """heavy_div (GDN_PRD_PUSH_DIV): 1 = every push with atomics, 10^9 = every push as a pull of the frontier's terms."""
    monkeypatch.setenv("GDN_PRD_PUSH_DIV", heavy_div)
    g = graphio.rmat_graph(scale, ef, seed=seed)
    gi = graphio.transpose(g)
    want, it, wtr = orc.pr_delta(gi, g, push_div=push_div)
    if push_div == 2:
        assert wtr["mode"].any()  # the push path is exercised
    r = solvers.ResidentPRDelta(solvers.Graph(csr=g, in_csr=gi))
    scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st, tr = r.run(scores, push_div=push_div)
    r.close()
    _prd_check(scores, tr, st, want, it, wtr)
    if heavy_div != "64" and wtr["mode"][:-1].any():
        pushes = tr["mode"] == 1
        assert np.array_equal(tr["masked"][pushes][:-1], np.full(pushes.sum() - 1, int(heavy_div != "1")))


@pytest.mark.parametrize("layout", [0, 1])
def test_pr_delta_layouts_and_reruns(orc, layout):
    """Both layouts of the pull's SpMV plan; a plan can be run again; pull-only runs repeat bit for bit."""
    g = graphio.rmat_graph(16, 16, seed=21)
    gi = graphio.transpose(g)
    want, it, wtr = orc.pr_delta(gi, g, push_div=8)
    r = solvers.ResidentPRDelta(solvers.Graph(csr=g, in_csr=gi), layout=layout)
    runs = []
    for _ in range(2):
        scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        st, tr = r.run(scores, push_div=8)
        _prd_check(scores, tr, st, want, it, wtr)
        runs.append((scores, tr))
    k = int(np.argmax(wtr["mode"])) if wtr["mode"].any() else it  # iterations before the first push
    assert np.array_equal(runs[0][1]["diff"][:k], runs[1][1]["diff"][:k])
    if not wtr["mode"].any():
        assert np.array_equal(runs[0][0], runs[1][0])
    # max_iter cuts the run like the reference's loop condition
    scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st, tr = r.run(scores, max_iter=3)
    w3, it3, _ = orc.pr_delta(gi, g, push_div=8, max_iter=3)
    assert st["iterations"] == it3 == 3
    np.testing.assert_allclose(scores, w3, rtol=REL_TOL, atol=0)
    r.close()


def test_pr_delta_degenerate_graphs(orc):
    """Vertices without out-edges (quotient by 0 in the reference, never read), without in-edges, an empty graph."""
    for g in [graphio.build_csr(5, np.array([0, 0, 1, 3], np.int64), np.array([1, 2, 2, 2], np.int64)),
              graphio.build_csr(4, np.zeros(0, np.int64), np.zeros(0, np.int64)),
              graphio.build_csr(70000, np.arange(69999, dtype=np.int64), np.arange(1, 70000, dtype=np.int64))]:
        gi = graphio.transpose(g)
        want, it, wtr = orc.pr_delta(gi, g, push_div=8)
        scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        st = solvers.PRDeltaSolver(solvers.Graph(csr=g, in_csr=gi), scores)
        assert st["iterations"] == it
        np.testing.assert_allclose(scores, want, rtol=REL_TOL, atol=0)


@pytest.mark.parametrize("tiers", ["0", "2"])
def test_spmv_pattern_plan_equals_unit_values(orc, monkeypatch, tiers):
    """GDN_LAYOUT_PB without values = the pattern matrix: the same sums as the plan that stores Ax = 1."""
    import ctypes as C
    from gardenia_amd import _cabi
    if tiers == "2":
        monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1")
    else:
        monkeypatch.setenv("GDN_PB_HUBS", "0")
    g = graphio.transpose(graphio.rmat_graph(16, 16, seed=31))
    rng = np.random.default_rng(3)
    x = (rng.random(g.m, dtype=np.float32) - np.float32(0.5)) * np.float32(3.0)
    L = _cabi.lib()
    h = C.c_void_p()
    rp, ci = np.ascontiguousarray(g.rowptr, np.uint64), np.ascontiguousarray(g.colidx, np.int32)
    _cabi.check(L.gdn_graph_upload(g.m, g.nnz, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), C.byref(h)))
    dA, dx, dy = C.c_void_p(), C.c_void_p(), C.c_void_p()
    ones = np.ones(g.nnz, np.float32)
    for d, a in ((dA, ones), (dx, x), (dy, np.zeros(g.m, np.float32))):
        _cabi.check(L.gdn_dev_alloc(a.nbytes, C.byref(d)))
        _cabi.check(L.gdn_dev_upload(d, a.ctypes.data_as(C.c_void_p), a.nbytes))
    outs = []
    for ax in (dA, None):
        plan = C.c_void_p()
        _cabi.check(L.gdn_spmv_plan_create(h, ax, _cabi.GDN_LAYOUT_PB, C.byref(plan)))
        y = np.zeros(g.m, np.float32)
        _cabi.check(L.gdn_dev_upload(dy, y.ctypes.data_as(C.c_void_p), y.nbytes))
        _cabi.check(L.gdn_spmv_dev(plan, ax, dx, dy, None))
        _cabi.check(L.gdn_spmv_plan_check(plan))
        _cabi.check(L.gdn_dev_download(y.ctypes.data_as(C.c_void_p), dy, y.nbytes))
        outs.append(y)
        L.gdn_spmv_plan_free(plan)
    for d in (dA, dx, dy):
        L.gdn_dev_free(d)
    L.gdn_graph_free(h)
    # the same bits -- except the few rows the VALUE plan recomputes in fp32 because a product lost bits in the fixed-point
    # conversion while the row's sum is tiny (signed x: cancellation); the pattern plan keeps its exact integer sums there
    same = outs[0] == outs[1]
    assert same.mean() > 0.99
    np.testing.assert_allclose(outs[0][~same], outs[1][~same], rtol=1e-4, atol=1e-6)
    want = orc.spmv(g, ones, x, np.zeros(g.m, np.float32))
    np.testing.assert_allclose(outs[1], want, rtol=REL_TOL, atol=1e-5)


@pytest.mark.parametrize("world", [1, 2, 5])
def test_tc_row_range_shards_add_up(orc, world):
    """The multi-GPU triangle count on one device: the DAG is oriented once (gdn_graph_orient), every 'rank' counts the
    rows of its edge-balanced range (gdn_tc_rows_dev), the partial counts add up to the oracle's total."""
    import ctypes as C
    from gardenia_amd import _cabi
    from gardenia_amd.sharded import HipTCBackend, ShardedTC
    g = graphio.symmetrize(graphio.rmat_graph(15, 16, seed=41))
    want = orc.tc(orc.tc_orient(g))
    L = _cabi.lib()
    h = C.c_void_p()
    rp, ci = np.ascontiguousarray(g.rowptr, np.uint64), np.ascontiguousarray(g.colidx, np.int32)
    _cabi.check(L.gdn_graph_upload(g.m, g.nnz, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), C.byref(h)))
    be = HipTCBackend(h, oriented=False, device=None)
    drp = be.rowptr()
    assert np.array_equal(drp, orc.tc_orient(g).rowptr)  # the device orientation is the reference's
    parts = [ShardedTC(be, drp, r, world, dist=None) for r in range(world)]
    assert parts[0].lo == 0 and parts[-1].hi == g.m and all(a.hi == b.lo for a, b in zip(parts, parts[1:]))
    got = [be.count_rows(p.lo, p.hi) for p in parts]
    assert sum(got) == want > 0
    if world > 1:
        assert max(got) < want  # a proper split
    assert be.count_rows(7, 7) == 0
    be.close()
    L.gdn_graph_free(h)


@pytest.mark.parametrize("small_nf", ["0", "256", "100000"])
def test_bfs_fused_light_levels(orc, monkeypatch, small_nf):
    """bfs_td_small_kernel (consecutive tiny levels inside one workgroup): off, default limits, and limits so wide that
    whole searches of small graphs run fused -- same depths and level counts as the serial BFS."""
    monkeypatch.setenv("GDN_BFS_SMALL_NF", small_nf)
    if small_nf == "100000":
        monkeypatch.setenv("GDN_BFS_SMALL_SCOUT", "100000000")
    m = 6000
    chain = graphio.build_csr(m, np.arange(m - 1, dtype=np.int64), np.arange(1, m, dtype=np.int64))
    cases = [(chain, 0), (chain, 5990), (graphio.rmat_graph(12, 8, seed=9), None), (graphio.rmat_graph(15, 16, seed=10), None),
             (graphio.symmetrize(graphio.rmat_graph(10, 4, seed=11)), None)]
    for g, src in cases:
        src = graphio.first_nonisolated(g) if src is None else src
        want = orc.bfs_serial(g, src)
        for G in (solvers.Graph(csr=g), solvers.Graph(csr=g, need_reverse=True)):
            d = np.full(g.m, 1000000000, np.int32)
            st = solvers.BFSSolver(G, src, d)
            assert np.array_equal(d, want)
            if g is chain and not G.has_reverse_graph():
                assert st["iterations"] == m - src  # one level per vertex, the last one discovers nothing


@pytest.mark.parametrize("small_nf", ["0", "256", "1000000"])
def test_bc_fused_light_levels(orc, monkeypatch, small_nf):
    """bc_fwd_small_kernel (consecutive tiny forward levels inside one workgroup): off, default limits, and limits so
    wide that whole forward phases run fused -- the reference verifier's criterion and the same level count."""
    monkeypatch.setenv("GDN_BC_SMALL_NF", small_nf)
    if small_nf == "1000000":
        monkeypatch.setenv("GDN_BC_SMALL_SCOUT", "100000000000")
    m = 3000
    chain = graphio.build_csr(m, np.arange(m - 1, dtype=np.int64), np.arange(1, m, dtype=np.int64))
    ladder = graphio.build_csr(m, np.concatenate([np.arange(m - 2), np.arange(m - 2)]).astype(np.int64),
                               np.concatenate([np.arange(1, m - 1), np.arange(2, m)]).astype(np.int64))
    for g, src in [(chain, 0), (ladder, 0), (graphio.rmat_graph(12, 8, seed=9), None),
                   (graphio.symmetrize(graphio.rmat_graph(11, 4, seed=11)), None), (graphio.rmat_graph(15, 16, seed=10), None)]:
        src = graphio.first_nonisolated(g) if src is None else src
        want, levels, _, _ = orc.bc(g, src)
        sc = np.zeros(g.m, np.float32)
        st = solvers.BCSolver(solvers.Graph(csr=g), src, sc)
        assert orc.bc_verify(g, src, sc)
        _bc_close(sc, want)
        assert st["iterations"] == levels


def test_bc_fused_backward_levels_same_bits(orc, monkeypatch):
    """bc_back_small_kernel (consecutive light backward levels inside one workgroup) off / default / every level that
    fits: the scores are the SAME BITS -- every row class is summed in one order whichever kernel runs its level --
    and satisfy the reference verifier.  The funnel graph puts a 6000-edge row (workgroup class) and 100-edge rows
    (wave class) into one-vertex levels."""
    m = 3000
    chain = graphio.build_csr(m, np.arange(m - 1, dtype=np.int64), np.arange(1, m, dtype=np.int64))
    hub, nleaf = 20, 6000
    src_l = list(range(hub)) + [hub] * nleaf
    dst_l = list(range(1, hub + 1)) + list(range(hub + 1, hub + 1 + nleaf))
    mid = hub + 1 + nleaf  # every leaf -> 100 collectors -> a tail chain
    for leaf in range(hub + 1, hub + 1 + nleaf):
        src_l += [leaf, leaf]
        dst_l += [mid + leaf % 100, mid + (leaf * 7) % 100]
    tail = mid + 100
    for c in range(100):
        src_l.append(mid + c)
        dst_l.append(tail)
    for t in range(50):
        src_l.append(tail + t)
        dst_l.append(tail + t + 1)
    funnel = graphio.build_csr(tail + 51, np.array(src_l, np.int64), np.array(dst_l, np.int64))
    cases = [(chain, 0), (funnel, 0), (graphio.symmetrize(graphio.rmat_graph(11, 4, seed=11)), None),
             (graphio.rmat_graph(15, 16, seed=10), None)]
    got = {}
    for mode in ("off", "default", "wide"):
        if mode == "off":
            monkeypatch.setenv("GDN_BC_BACK_NF", "0")
        elif mode == "wide":
            monkeypatch.setenv("GDN_BC_BACK_NF", "1024")
            monkeypatch.setenv("GDN_BC_BACK_SCOUT", "100000000000")
        else:
            monkeypatch.delenv("GDN_BC_BACK_NF", raising=False)
        for k, (g, src) in enumerate(cases):
            src = graphio.first_nonisolated(g) if src is None else src
            sc = np.zeros(g.m, np.float32)
            st = solvers.BCSolver(solvers.Graph(csr=g), src, sc)
            assert orc.bc_verify(g, src, sc), (mode, k)
            got[(mode, k)] = sc
    for k in range(len(cases)):
        for mode in ("default", "wide"):
            assert np.array_equal(got[("off", k)].view(np.uint32), got[(mode, k)].view(np.uint32)), (mode, k)



def test_pr_fused_solve(orc, monkeypatch):
    """pr_fused_kernel against the per-iteration loop and the oracle: the same iteration count, every line of the trace,
    scores within 1e-4 (short rows: the reference's own summation order), the same bits run after run, max_iter honoured,
    hub rows (wave class) and vertices without in-edges included."""
    monkeypatch.delenv("GDN_PR_LAYOUT", raising=False)
    # (a 300-leaf star: the reference's sequential fp32 sum of n equal terms is itself off by ~n/2 ulp, 1.6e-4 at n = 3000)
    star_src = np.concatenate([np.arange(1, 300), np.zeros(40, np.int64)]).astype(np.int64)
    star_dst = np.concatenate([np.zeros(299, np.int64), np.arange(1, 41)]).astype(np.int64)
    graphs = [graphio.rmat_graph(9, 4, seed=3), graphio.rmat_graph(14, 16, seed=5), graphio.rmat_graph(12, 64, seed=7),
              graphio.build_csr(3000, star_src, star_dst)]
    for gk, g in enumerate(graphs):
        gi = graphio.transpose(g)
        G = solvers.Graph(csr=g, in_csr=gi)
        want, it, trace = orc.pr(gi, g.degrees())
        res = {}
        for mode in ("0", "1", "1"):
            monkeypatch.setenv("GDN_PR_FUSED", mode)
            s = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
            st = solvers.PRSolver(G, s)
            assert st["iterations"] == it, (gk, mode)
            np.testing.assert_allclose(s, want, rtol=REL_TOL, atol=0, err_msg="graph %d mode %s" % (gk, mode))
            assert len(st["trace"]) == len(trace)
            np.testing.assert_allclose(st["trace"], trace, rtol=1e-3, atol=1e-9)
            assert abs(st["last_error"] - st["trace"][-1]) == 0
            if mode == "1" and "1" in res:
                assert np.array_equal(res["1"].view(np.uint32), s.view(np.uint32))
            res[mode] = s
        np.testing.assert_allclose(res["1"], res["0"], rtol=2e-5, atol=0)
        s = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        st = solvers.PRSolver(G, s, max_iter=3)
        assert st["iterations"] in (3, 4) and len(st["trace"]) == 3  # the loop ran 3 times without converging
    # the one-workgroup form on a graph the grid form would take by itself, and the other way round
    g = graphio.rmat_graph(13, 8, seed=21)
    gi = graphio.transpose(g)
    want, it, _ = orc.pr(gi, g.degrees())
    monkeypatch.setenv("GDN_PR_FUSED", "1")
    for small_m in ("16384", "0"):
        monkeypatch.setenv("GDN_PR_SMALL_M", small_m)
        s = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        st = solvers.PRSolver(solvers.Graph(csr=g, in_csr=gi), s)
        assert st["iterations"] == it
        np.testing.assert_allclose(s, want, rtol=REL_TOL, atol=0)


@pytest.mark.parametrize("layout", ["csr", "pb"])
def test_pr_batched_loop_stops_where_the_per_iteration_loop_stops(orc, monkeypatch, layout):
    """gdn_pr queues its iterations in batches and tests convergence on the device (pr_check_kernel + GdnSkippable): the
    scores are the bits of the per-iteration loop (GDN_PR_BATCH=1) for every batch size, the trace and the iteration
    count are the oracle's, and max_iter is honoured inside a batch."""
    monkeypatch.setenv("GDN_PR_LAYOUT", layout)
    monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1")
    for g in (graphio.rmat_graph(13, 8, seed=2), graphio.rmat_graph(16, 16, seed=3)):
        gi = graphio.transpose(g)
        G = solvers.Graph(csr=g, in_csr=gi)
        want, it, trace = orc.pr(gi, g.degrees())
        ref = None
        for batch in ("1", "3", "8", "64"):
            monkeypatch.setenv("GDN_PR_BATCH", batch)
            s = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
            st = solvers.PRSolver(G, s)
            assert st["iterations"] == it and len(st["trace"]) == len(trace)
            np.testing.assert_allclose(st["trace"], trace, rtol=1e-3, atol=1e-9)
            np.testing.assert_allclose(s, want, rtol=REL_TOL, atol=0)
            if ref is None:
                ref = (s, st["trace"])
            else:
                assert np.array_equal(ref[0].view(np.uint32), s.view(np.uint32))
                assert np.array_equal(np.asarray(ref[1]), np.asarray(st["trace"]))
        for batch in ("1", "8"):
            monkeypatch.setenv("GDN_PR_BATCH", batch)
            s = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
            st = solvers.PRSolver(G, s, max_iter=5)
            assert len(st["trace"]) == 5 and st["iterations"] == 6


@pytest.mark.parametrize("mode", ["forced", "forced_early", "overflow"])
def test_bfs_binned_top_down_level(orc, monkeypatch, mode):
    """The binned top-down level (bfs_btd_*: the frontier's out-edges binned by destination range, then applied per bin in
    LDS) forced onto graphs far below the size at which the plan picks it: exact depths against the oracle; `forced_early`
    also lets the bitmap phase start for small frontiers; `overflow`: lists far too short for a frontier whose destinations
    are not spread over the bins (every edge of a star lands in a few bins) -- the flag sends the level to the sweep."""
    monkeypatch.setenv("GDN_BFS_BTD", "2")
    if mode == "forced_early":
        monkeypatch.setenv("GDN_BFS_ALPHA_BTD", "100000")
        monkeypatch.setenv("GDN_BFS_BTD_MIN", "1")
    graphs = [graphio.rmat_graph(16, 16, seed=3), graphio.symmetrize(graphio.rmat_graph(15, 8, seed=4)),
              graphio.rmat_graph(18, 8, seed=5)]
    if mode == "overflow":
        m = 1 << 17  # 8 bins of 16384 ids; vertex 0 points at 16000 ids of bin 0 alone, whose 8 lists hold ~10 K of them
        hub_dst = np.arange(1, 16001, dtype=np.int64)
        src = np.concatenate([np.zeros(16000, np.int64), hub_dst, np.arange(16001, m - 1)])
        dst = np.concatenate([hub_dst, (hub_dst * 7919) % m, np.arange(16002, m)])
        graphs = [graphio.build_csr(m, src, dst)]
    for g in graphs:
        gi = graphio.transpose(g)
        G = solvers.Graph(csr=g, in_csr=gi)
        bfs = solvers.ResidentBFS(G, dense=True)
        for s in ([0] if mode == "overflow" else [graphio.first_nonisolated(g), int(np.argmax(g.degrees()))]):
            want = orc.bfs_serial(g, s)
            d, st = bfs.run(s)
            assert np.array_equal(d, want), (mode, s)
        bfs.close()
