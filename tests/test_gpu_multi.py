"""The multi-GPU drop-in entries of the C-ABI (gdn_pr_multi / gdn_spmv_multi: one process, one host thread per device)
on ONE device -- listing device 0 several times makes the ranks share it and the slices travel by peer copies (RCCL
refuses duplicate devices).  With the propagation-blocked layout the scores must equal the single-device solver's BIT FOR
BIT for every rank count (integer accumulation); the merge-path layout agrees to 1e-6; both match the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from gardenia_amd import _cabi, graphio, solvers

pytestmark = pytest.mark.gpu


def _pr_graph(scale=15, ef=16, seed=41, cut=5):
    g = graphio.rmat_graph(scale, ef, seed=seed)
    m = g.m - cut  # not divisible by the rank count
    src, dst = graphio.csr_to_coo(g)
    keep = (src < m) & (dst < m)
    g = graphio.build_csr(m, src[keep], dst[keep])
    return g, graphio.transpose(g)


@pytest.mark.parametrize("ranks,parts", [(2, "1"), (3, "3"), (5, "4"), (2, "8")])
def test_pr_multi_pb_bits_equal_single_device(orc, monkeypatch, ranks, parts):
    monkeypatch.setenv("GDN_PR_LAYOUT", "pb")
    monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1")  # record tiers on the shards too
    # (round 6: the exchange pipelined in `parts` row ranges of ONE ticketed launch per phase, gdn_pr_pull_parts_dev: every
    # part's peer copies are queued behind the kernel that waits for its tickets)
    monkeypatch.setenv("GDN_MULTI_PARTS", parts)
    g, gi = _pr_graph()
    G = solvers.Graph(csr=g, in_csr=gi)
    one = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st1 = solvers.PRSolver(G, one)
    many = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    stn = solvers.PRSolver(G, many, devices=[0] * ranks)
    assert stn["reserved"] == 2  # peer copies (the ranks share a device)
    assert stn["iterations"] == st1["iterations"]
    assert np.array_equal(one.view(np.uint32), many.view(np.uint32))
    np.testing.assert_allclose(stn["trace"], st1["trace"], rtol=1e-9)
    want, it, trace = orc.pr(gi, g.degrees())
    assert it == stn["iterations"]
    np.testing.assert_allclose(many, want, rtol=1e-4, atol=0)
    np.testing.assert_allclose(stn["trace"], trace, rtol=1e-3)


@pytest.mark.parametrize("ranks", [1, 2, 4])
def test_pr_multi_default_layout_vs_oracle(orc, ranks):
    g, gi = _pr_graph(13, 16, 42, 3)
    G = solvers.Graph(csr=g, in_csr=gi)
    want, it, trace = orc.pr(gi, g.degrees())
    scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st = solvers.PRSolver(G, scores, devices=[0] * ranks)
    assert st["iterations"] == it and len(st["trace"]) == it
    np.testing.assert_allclose(scores, want, rtol=1e-4, atol=0)
    assert abs(st["last_error"] - trace[-1]) < 1e-6
    assert orc.pr_verify_error(g, scores) < 1e-4


def test_pr_multi_max_iter_limited_reports_like_gdn_pr(monkeypatch):
    """A solve cut off by max_iter reports MAX_ITER + 1 iterations whatever GDN_NUM_GPUS is, like the reference prints
    iter + 1 after the loop (src/pr/omp_base.cc:39); a malformed GDN_MULTI_DEVICES is an error, not a hang."""
    g, gi = _pr_graph(13, 16, 46, 3)
    G = solvers.Graph(csr=g, in_csr=gi)
    a = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    b = a.copy()
    st1 = solvers.PRSolver(G, a, max_iter=3)
    st2 = solvers.PRSolver(G, b, max_iter=3, devices=[0, 0])
    assert st1["iterations"] == st2["iterations"] == 4
    np.testing.assert_allclose(a, b, rtol=1e-6)
    for bad in ("0;0", "0,x", "0,0\n1"):
        monkeypatch.setenv("GDN_MULTI_DEVICES", bad)
        with pytest.raises(_cabi.GardeniaError) as ei:
            solvers.PRSolver(G, b, ngpus=2)
        assert ei.value.status == _cabi.GDN_ERR_INVALID and "GDN_MULTI_DEVICES" in str(ei.value)
    monkeypatch.setenv("GDN_MULTI_DEVICES", "0, 0")
    st3 = solvers.PRSolver(G, a.copy(), max_iter=3, ngpus=2)
    assert st3["iterations"] == 4


def test_pr_multi_more_ranks_than_rows_and_bad_device():
    g = graphio.build_csr(3, np.array([0, 1, 2]), np.array([1, 2, 0]))
    gi = graphio.transpose(g)
    G = solvers.Graph(csr=g, in_csr=gi)
    scores = np.full(3, np.float32(1.0 / 3.0), np.float32)
    st = solvers.PRSolver(G, scores, devices=[0] * 8)  # clamped to 3 ranks
    np.testing.assert_allclose(scores, 1.0 / 3.0, rtol=1e-6)
    assert st["iterations"] >= 1
    with pytest.raises(_cabi.GardeniaError) as ei:
        solvers.PRSolver(G, scores, devices=[0, 63])
    assert ei.value.status == _cabi.GDN_ERR_INVALID and "does not exist" in str(ei.value)


def test_pr_multi_rccl_single_rank(orc, monkeypatch):
    """GDN_MULTI_EXCHANGE=rccl drives librccl (dlopen, ncclCommInitAll, in-place ncclAllGather) with the one rank a
    1-GPU box has."""
    monkeypatch.setenv("GDN_MULTI_EXCHANGE", "rccl")
    g, gi = _pr_graph(12, 16, 43, 1)
    want, it, _ = orc.pr(gi, g.degrees())
    scores = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st = solvers.PRSolver(solvers.Graph(csr=g, in_csr=gi), scores, devices=[0])
    assert st["reserved"] == 1 and st["iterations"] == it
    np.testing.assert_allclose(scores, want, rtol=1e-4, atol=0)


@pytest.mark.parametrize("ranks", [2, 3])
def test_spmv_multi_bits_equal_single_device(orc, ranks):
    g = graphio.rmat_graph(14, 16, seed=44)
    gi = graphio.transpose(g)
    G = solvers.Graph(csr=g, in_csr=gi)
    rng = np.random.default_rng(9)
    Ax, x, y0 = (rng.random(n).astype(np.float32) for n in (g.nnz, g.m, g.m))
    y1, yn = y0.copy(), y0.copy()
    solvers.SpmvSolver(G, Ax, x, y1)
    solvers.SpmvSolver(G, Ax, x, yn, devices=[0] * ranks)
    want = orc.spmv(gi, Ax, x, y0)
    np.testing.assert_allclose(yn, want, rtol=1e-4, atol=0)
    assert orc.spmv_max_rel_error(yn, want) <= 5 * np.sqrt(np.finfo(np.float32).eps)
    np.testing.assert_allclose(yn, y1, rtol=1e-6, atol=0)


def test_multi_ranges_follow_the_edges():
    g = graphio.rmat_graph(14, 16, seed=45)  # natural (unpermuted-looking) skew: equal vertex counts are not equal work
    L = _cabi.lib()
    for w in (2, 3, 8):
        b, ch = (C.c_int32 * (w + 1))(), C.c_int32(0)
        rp = np.ascontiguousarray(g.rowptr, np.uint64)
        _cabi.check(L.gdn_multi_ranges(g.m, rp.ctypes.data_as(C.c_void_p), w, b, C.byref(ch)))
        b = np.array(list(b))
        assert b[0] == 0 and b[-1] == g.m and np.all(np.diff(b) > 0)
        per = np.diff(g.rowptr[b].astype(np.int64))
        assert per.max() <= g.nnz / w + g.degrees().max() + 1
        assert ch.value >= np.diff(b).max() and ch.value % 4 == 0
        # the resident-graph form gives the same cut, and its padded slice relabels the columns like the numpy mirror
        h = C.c_void_p()
        ci = np.ascontiguousarray(g.colidx, np.int32)
        _cabi.check(L.gdn_graph_upload(g.m, g.nnz, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), C.byref(h)))
        b2 = (C.c_int32 * (w + 1))()
        _cabi.check(L.gdn_graph_balanced_ranges(h, w, b2))
        assert list(b2) == b.tolist()
        from gardenia_amd.sharded import pad_columns
        for r in (0, w - 1):
            sh = C.c_void_p()
            _cabi.check(L.gdn_graph_slice_padded(h, w, b2, ch.value, r, C.byref(sh)))
            sm, sn = C.c_int32(), C.c_uint64()
            _cabi.check(L.gdn_graph_info(sh, C.byref(sm), C.byref(sn), None, None))
            got_rp, got_ci = np.empty(sm.value + 1, np.uint64), np.empty(sn.value, np.int32)
            _cabi.check(L.gdn_graph_download(sh, got_rp.ctypes.data_as(C.c_void_p), got_ci.ctypes.data_as(C.c_void_p)))
            e0, e1 = int(g.rowptr[b[r]]), int(g.rowptr[b[r + 1]])
            assert np.array_equal(got_rp, g.rowptr[b[r]:b[r + 1] + 1] - g.rowptr[b[r]])
            assert np.array_equal(got_ci, pad_columns(g.colidx[e0:e1], b, ch.value))
            L.gdn_graph_free(sh)
        L.gdn_graph_free(h)


def test_graph_upload_rejects_malformed_csr():
    L = _cabi.lib()
    rp = np.array([0, 2, 3, 4], np.uint64)
    h = C.c_void_p()
    bad_col = np.array([1, 2, 7, 0], np.int32)  # column id outside [0, m)
    assert L.gdn_graph_upload(3, 4, rp.ctypes.data_as(C.c_void_p), bad_col.ctypes.data_as(C.c_void_p), C.byref(h)) == _cabi.GDN_ERR_INVALID
    assert b"column id" in L.gdn_last_error()
    neg = np.array([1, -2, 0, 0], np.int32)
    assert L.gdn_graph_upload(3, 4, rp.ctypes.data_as(C.c_void_p), neg.ctypes.data_as(C.c_void_p), C.byref(h)) == _cabi.GDN_ERR_INVALID
    bad_rp = np.array([0, 3, 2, 4], np.uint64)  # not ascending
    ok_col = np.array([1, 2, 0, 0], np.int32)
    assert L.gdn_graph_upload(3, 4, bad_rp.ctypes.data_as(C.c_void_p), ok_col.ctypes.data_as(C.c_void_p), C.byref(h)) == _cabi.GDN_ERR_INVALID
    assert b"offsets" in L.gdn_last_error()
    assert L.gdn_graph_upload(3, 4, rp.ctypes.data_as(C.c_void_p), ok_col.ctypes.data_as(C.c_void_p), C.byref(h)) == _cabi.GDN_OK
    assert L.gdn_graph_validate(h, 3) == _cabi.GDN_OK and L.gdn_graph_validate(h, 2) == _cabi.GDN_ERR_INVALID
    L.gdn_graph_free(h)


def test_pr_trace_matches_the_reference_line_by_line():
    """All 15 lines of the only golden the reference ships (test/reference/graph-pr.mtx.out:13-27), at its print
    precision, from the ctypes path (gdn_pr_last_trace) and from the pr_hip main (printed like src/pr/omp_base.cc:35),
    on one device and on two ranks."""
    import json
    gold = json.load(open(os.path.join(GOLDEN, "pr_trace_golden.json")))
    want = ["%.6f" % v for v in gold["trace"]]
    assert len(want) == gold["iterations"] == 15
    g = solvers.Graph(os.path.join(GOLDEN, "graphs", "test_pr"), "mtx", False, True)
    for devs in (None, [0, 0]):
        scores = np.full(g.V(), np.float32(1.0) / np.float32(g.V()), np.float32)
        st = solvers.PRSolver(g, scores, devices=devs)
        assert st["iterations"] == 15
        assert ["%.6f" % v for v in st["trace"]] == want
    exe = os.path.join(ROOT, "gardenia_amd", "host", "bin", "pr_hip")
    for env in ({}, {"GDN_NUM_GPUS": "2", "GDN_MULTI_DEVICES": "0,0"}):
        p = subprocess.run([exe, "mtx", os.path.join(GOLDEN, "graphs", "test_pr"), "0"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=300, env=dict(os.environ, **env))
        assert p.returncode == 0 and "Correct" in p.stdout, p.stdout[-800:]
        lines = [ln for ln in p.stdout.splitlines() if len(ln.split()) == 2 and ln.split()[0].isdigit()]
        assert lines == [" %2d    %s" % (i + 1, w) for i, w in enumerate(want)], p.stdout
        assert "iterations = 15." in p.stdout
        assert ("2 GPUs" in p.stdout) == bool(env)


def test_spmv_main_on_two_ranks(tmp_path):
    g = graphio.rmat_graph(13, 16, seed=46)
    graphio.write_bin(str(tmp_path / "rm"), g)
    exe = os.path.join(ROOT, "gardenia_amd", "host", "bin", "spmv_hip")
    p = subprocess.run([exe, "bin", str(tmp_path / "rm"), "0", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=300, env=dict(os.environ, GDN_NUM_GPUS="3", GDN_MULTI_DEVICES="0,0,0"))
    assert p.returncode == 0 and "Correct" in p.stdout, p.stdout[-800:]


def test_option_set_through_the_api_picks_the_layout(monkeypatch):
    """gdn_option_set("GDN_PR_LAYOUT", ...) steers the AUTO layout of a plan like the environment variable does."""
    monkeypatch.delenv("GDN_PR_LAYOUT", raising=False)
    L = _cabi.lib()
    g = graphio.transpose(graphio.rmat_graph(14, 16, seed=3))
    rp, ci = np.ascontiguousarray(g.rowptr, np.uint64), np.ascontiguousarray(g.colidx, np.int32)
    h, d_deg = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_graph_upload(g.m, g.nnz, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), C.byref(h)))
    _cabi.check(L.gdn_dev_alloc(4 * g.m, C.byref(d_deg)))
    _cabi.check(L.gdn_graph_degrees_dev(h, d_deg, None))
    got = {}
    try:
        for val in (None, b"pb", b"csr"):
            _cabi.check(L.gdn_option_set(b"GDN_PR_LAYOUT", val))
            plan, lay = C.c_void_p(), C.c_int32(-9)
            _cabi.check(L.gdn_pr_plan_create(h, d_deg, g.m, 0, _cabi.GDN_LAYOUT_AUTO, C.byref(plan)))
            _cabi.check(L.gdn_pr_plan_layout(plan, C.byref(lay), None))
            L.gdn_pr_plan_free(plan)
            got[val] = lay.value
    finally:
        L.gdn_option_set(b"GDN_PR_LAYOUT", None)
        L.gdn_dev_free(d_deg)
        L.gdn_graph_free(h)
    assert got == {None: _cabi.GDN_LAYOUT_CSR, b"pb": _cabi.GDN_LAYOUT_PB, b"csr": _cabi.GDN_LAYOUT_CSR}  # 262 K edges: CSR by size
