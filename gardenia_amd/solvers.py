"""Host-side mirror of the reference's per-kernel Solver API on numpy arrays.

Same names, argument meaning and in/out conventions as the reference harness
(src/<kernel>/main.cc + <kernel>.h); each function is one call through the C-ABI of
include/gardenia_hip.h into the HIP kernels.  `Graph` mirrors the accessor surface of
include/csr_graph.h:265-306 that the solvers use.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _cabi
from .graphio import CSR, read_bin, read_mtx, transpose

MYINFINITY = 1000000000   # include/common.h:66
K_DIST_INF = 2147483647   # src/sssp/sssp.h:46 (UINT_MAX/2 as int)
K_DAMP = np.float32(0.85) # src/pr/pr.h:6
EPSILON = 0.0001          # src/pr/pr.h:5
MAX_ITER = 100            # src/pr/pr.h:12


class Graph:
    """include/csr_graph.h Graph: (prefix, filetype, symmetrize, need_reverse) ctor :211-250."""

    def __init__(self, prefix: Optional[str] = None, filetype: str = "bin", symmetrize: bool = False,
                 need_reverse: bool = False, csr: Optional[CSR] = None, in_csr: Optional[CSR] = None):
        if csr is None:
            if filetype == "mtx":
                csr = read_mtx(prefix + ".mtx", symmetrize)
            elif filetype == "bin":
                csr = read_bin(prefix)
            else:
                raise ValueError("filetype must be 'mtx' or 'bin'")
        self._out = csr
        # directed = the reverse graph is a graph of its own (built here, or handed in as in_csr)
        self._directed = (not symmetrize) and (need_reverse or (in_csr is not None and in_csr is not csr))
        self._in = None
        if in_csr is not None:
            self._in = in_csr
        elif symmetrize:
            self._in = csr  # csr_graph.h:241-245: reverse_* alias the forward arrays
        elif need_reverse:
            self._in = transpose(csr)

    def V(self): return self._out.m
    def E(self): return self._out.nnz
    def out_rowptr(self): return self._out.rowptr
    def out_colidx(self): return self._out.colidx
    def has_reverse_graph(self): return self._in is not None
    def is_directed(self): return self._directed
    def in_rowptr(self): return self._in.rowptr
    def in_colidx(self): return self._in.colidx
    def get_degree(self, v): return int(self._out.rowptr[v + 1] - self._out.rowptr[v])
    def out_degrees(self): return self._out.degrees()
    def out_csr(self): return self._out
    def in_csr(self): return self._in


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def BFSSolver(g: Graph, source: int, dist: np.ndarray) -> dict:
    """src/bfs/bfs.h:43.  dist: int32[m], pre-filled with MYINFINITY by the caller."""
    assert dist.dtype == np.int32 and dist.flags.c_contiguous
    rp, ci = _arr(g.out_rowptr(), np.uint64), _arr(g.out_colidx(), np.int32)
    irp = ici = None
    if g.has_reverse_graph():
        irp, ici = _arr(g.in_rowptr(), np.uint64), _arr(g.in_colidx(), np.int32)
    st = _cabi.GdnStats()
    _cabi.check(_cabi.lib().gdn_bfs(g.V(), g.E(), _p(rp), _p(ci), _p(irp), _p(ici), source, _p(dist), C.byref(st)))
    return st.as_dict()


def num_gpus() -> int:
    """GDN_NUM_GPUS: how many devices PRSolver / SpmvSolver spread over (the reference's OMP_NUM_THREADS analogue)."""
    import os
    try:
        return max(1, int(os.environ.get("GDN_NUM_GPUS", "1")))
    except ValueError:
        return 1


def pr_last_trace() -> np.ndarray:
    """L1 change of every iteration of this thread's last PRSolver call (what src/pr/omp_base.cc:35 prints)."""
    n = C.c_int32(0)
    _cabi.check(_cabi.lib().gdn_pr_last_trace(0, C.byref(n), None))
    d = np.zeros(max(n.value, 1), np.float64)
    _cabi.check(_cabi.lib().gdn_pr_last_trace(n.value, C.byref(n), _p(d)))
    return d[:n.value]


def PRSolver(g: Graph, scores: np.ndarray, damping=K_DAMP, epsilon=EPSILON, max_iter=MAX_ITER, ngpus: Optional[int] = None,
             devices=None) -> dict:
    """src/pr/pr.h:31.  scores: float32[m], pre-filled with 1/m by the caller.  ngpus (default GDN_NUM_GPUS or 1) > 1
    or an explicit `devices` list: gdn_pr_multi, vertex-range shards on that many devices."""
    assert scores.dtype == np.float32 and scores.flags.c_contiguous
    irp, ici = _arr(g.in_rowptr(), np.uint64), _arr(g.in_colidx(), np.int32)
    deg = _arr(g.out_degrees(), np.int32)
    st = _cabi.GdnStats()
    n = num_gpus() if ngpus is None else int(ngpus)
    if devices is not None or n > 1:
        dv = None if devices is None else _arr(devices, np.int32)
        n = n if dv is None else len(dv)
        _cabi.check(_cabi.lib().gdn_pr_multi(g.V(), g.E(), _p(irp), _p(ici), _p(deg), _p(scores), float(damping),
                                             float(epsilon), int(max_iter), n, _p(dv), C.byref(st)))
    else:
        _cabi.check(_cabi.lib().gdn_pr(g.V(), g.E(), _p(irp), _p(ici), _p(deg), _p(scores), float(damping),
                                       float(epsilon), int(max_iter), C.byref(st)))
    out = st.as_dict()
    out["trace"] = pr_last_trace()
    if not (devices is not None or n > 1):
        lay = C.c_int32(-1)
        _cabi.check(_cabi.lib().gdn_pr_last_layout(C.byref(lay)))
        out["layout"] = {0: "csr", 1: "pb"}.get(lay.value, str(lay.value))
    return out


def PRDeltaSolver(g: Graph, scores: np.ndarray, damping: float = 0.85, epsilon: float = 1e-4, epsilon2: float = 1e-3,
                  max_iter: int = 100, push_div: int = 8) -> dict:
    """The delta-PageRank PRSolver of src/pr/delta.cu:140 (push_div 8) / src/pr/omp_delta.cc:52 (push_div 10).
    scores: float32[m] pre-filled with 1/m (src/pr/main.cc:17); needs the reverse graph."""
    assert scores.dtype == np.float32 and scores.flags.c_contiguous
    irp, ici = _arr(g.in_rowptr(), np.uint64), _arr(g.in_colidx(), np.int32)
    orp, oci = _arr(g.out_rowptr(), np.uint64), _arr(g.out_colidx(), np.int32)
    st = _cabi.GdnStats()
    _cabi.check(_cabi.lib().gdn_pr_delta(g.V(), g.E(), _p(irp), _p(ici), _p(orp), _p(oci), _p(scores), damping, epsilon,
                                         epsilon2, max_iter, push_div, C.byref(st)))
    return st.as_dict()


def BCSolver(g: Graph, source: int, scores: np.ndarray) -> dict:
    """src/bc/bc.h:37.  scores: float32[m], zero-filled by the caller (src/bc/main.cc:21); one source."""
    assert scores.dtype == np.float32 and scores.flags.c_contiguous
    rp, ci = _arr(g.out_rowptr(), np.uint64), _arr(g.out_colidx(), np.int32)
    st = _cabi.GdnStats()
    _cabi.check(_cabi.lib().gdn_bc(g.V(), g.E(), _p(rp), _p(ci), int(source), _p(scores), C.byref(st)))
    return st.as_dict()


def SpmvSolver(g: Graph, Ax: np.ndarray, x: np.ndarray, y: np.ndarray, ngpus: Optional[int] = None, devices=None) -> dict:
    """src/spmv/spmv.h:29.  y += A x over the rows of (in_rowptr, in_colidx).  ngpus / devices as for PRSolver."""
    assert y.dtype == np.float32 and y.flags.c_contiguous
    irp, ici = _arr(g.in_rowptr(), np.uint64), _arr(g.in_colidx(), np.int32)
    Ax, x = _arr(Ax, np.float32), _arr(x, np.float32)
    st = _cabi.GdnStats()
    n = num_gpus() if ngpus is None else int(ngpus)
    if devices is not None or n > 1:
        dv = None if devices is None else _arr(devices, np.int32)
        n = n if dv is None else len(dv)
        _cabi.check(_cabi.lib().gdn_spmv_multi(g.V(), g.E(), _p(irp), _p(ici), _p(Ax), _p(x), _p(y), n, _p(dv), C.byref(st)))
    else:
        _cabi.check(_cabi.lib().gdn_spmv(g.V(), g.E(), _p(irp), _p(ici), _p(Ax), _p(x), _p(y), C.byref(st)))
    return st.as_dict()


def SSSPSolver(g: Graph, source: int, weight: np.ndarray, dist: np.ndarray, delta: int = 1) -> dict:
    """src/sssp/sssp.h:47.  dist: int32[m] pre-filled with kDistInf."""
    assert dist.dtype == np.int32 and dist.flags.c_contiguous
    rp, ci = _arr(g.out_rowptr(), np.uint64), _arr(g.out_colidx(), np.int32)
    w = _arr(weight, np.int32)
    st = _cabi.GdnStats()
    _cabi.check(_cabi.lib().gdn_sssp(g.V(), g.E(), _p(rp), _p(ci), _p(w), source, delta, _p(dist), C.byref(st)))
    return st.as_dict()


def TCSolver(g: Graph, oriented: bool = False):
    """src/tc/tc.h:7.  Returns (total, stats); orientation (USE_DAG) is applied unless oriented."""
    rp, ci = _arr(g.out_rowptr(), np.uint64), _arr(g.out_colidx(), np.int32)
    total = C.c_uint64(0)
    st = _cabi.GdnStats()
    _cabi.check(_cabi.lib().gdn_tc(g.V(), g.E(), _p(rp), _p(ci), 1 if oriented else 0, C.byref(total), C.byref(st)))
    return int(total.value), st.as_dict()


def CCSolver(g: Graph, comp: np.ndarray) -> dict:
    """src/cc/cc.h:28.  comp: int32[m] pre-filled with comp[i] = i."""
    assert comp.dtype == np.int32 and comp.flags.c_contiguous
    rp, ci = _arr(g.out_rowptr(), np.uint64), _arr(g.out_colidx(), np.int32)
    irp = ici = None
    if g.has_reverse_graph():  # directed: the reverse graph; symmetrized: the graph itself (alias)
        if g.is_directed():
            irp, ici = _arr(g.in_rowptr(), np.uint64), _arr(g.in_colidx(), np.int32)
        else:
            irp, ici = rp, ci
    st = _cabi.GdnStats()
    _cabi.check(_cabi.lib().gdn_cc(g.V(), g.E(), _p(rp), _p(ci), _p(irp), _p(ici), _p(comp), C.byref(st)))
    return st.as_dict()


class ResidentBFS:
    """Many BFS runs on one resident graph (gdn_bfs_plan_*): upload once, search from any source.
    dense=True also builds the propagation-blocked in-edge layout used for the heavy levels."""

    def __init__(self, g: Graph, dense: bool = True):
        L = _cabi.lib()
        self.L, self.m = L, g.V()
        self.h_out, self.h_in, self.plan, self.d_dist = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        rp, ci = _arr(g.out_rowptr(), np.uint64), _arr(g.out_colidx(), np.int32)
        _cabi.check(L.gdn_graph_upload(g.V(), g.E(), _p(rp), _p(ci), C.byref(self.h_out)))
        hin = None
        if g.has_reverse_graph():
            irp, ici = _arr(g.in_rowptr(), np.uint64), _arr(g.in_colidx(), np.int32)
            _cabi.check(L.gdn_graph_upload(g.V(), g.E(), _p(irp), _p(ici), C.byref(self.h_in)))
            hin = self.h_in
        _cabi.check(L.gdn_bfs_plan_create(self.h_out, hin, 1 if (dense and hin is not None) else 0, C.byref(self.plan)))
        _cabi.check(L.gdn_dev_alloc(4 * g.V(), C.byref(self.d_dist)))

    def run(self, source: int):
        st = _cabi.GdnStats()
        _cabi.check(self.L.gdn_bfs_run(self.plan, source, self.d_dist, C.byref(st)))
        dist = np.empty(self.m, np.int32)
        _cabi.check(self.L.gdn_dev_download(_p(dist), self.d_dist, 4 * self.m))
        return dist, st.as_dict()

    def close(self):
        self.L.gdn_bfs_plan_free(self.plan)
        self.L.gdn_dev_free(self.d_dist)
        self.L.gdn_graph_free(self.h_out)
        if self.h_in:
            self.L.gdn_graph_free(self.h_in)


class ResidentSpMV:
    """y += A x with the matrix resident (gdn_spmv_plan_*).  layout: 0 CSR merge-path, 1 PB."""

    def __init__(self, g: Graph, Ax: np.ndarray, layout: int = _cabi.GDN_LAYOUT_AUTO):
        L = _cabi.lib()
        self.L, self.m, self.nnz = L, g.V(), g.E()
        self.h, self.plan = C.c_void_p(), C.c_void_p()
        irp, ici = _arr(g.in_rowptr(), np.uint64), _arr(g.in_colidx(), np.int32)
        _cabi.check(L.gdn_graph_upload(g.V(), g.E(), _p(irp), _p(ici), C.byref(self.h)))
        self.d_Ax, self.d_x, self.d_y = C.c_void_p(), C.c_void_p(), C.c_void_p()
        Ax = _arr(Ax, np.float32)
        _cabi.check(L.gdn_dev_alloc(4 * max(self.nnz, 1), C.byref(self.d_Ax)))
        _cabi.check(L.gdn_dev_alloc(4 * self.m, C.byref(self.d_x)))
        _cabi.check(L.gdn_dev_alloc(4 * self.m, C.byref(self.d_y)))
        if self.nnz:
            _cabi.check(L.gdn_dev_upload(self.d_Ax, _p(Ax), 4 * self.nnz))
        _cabi.check(L.gdn_spmv_plan_create(self.h, self.d_Ax, layout, C.byref(self.plan)))

    def multiply(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        x, y = _arr(x, np.float32), np.array(y, dtype=np.float32)
        _cabi.check(self.L.gdn_dev_upload(self.d_x, _p(x), 4 * self.m))
        _cabi.check(self.L.gdn_dev_upload(self.d_y, _p(y), 4 * self.m))
        _cabi.check(self.L.gdn_spmv_dev(self.plan, self.d_Ax, self.d_x, self.d_y, None))
        _cabi.check(self.L.gdn_dev_download(_p(y), self.d_y, 4 * self.m))
        _cabi.check(self.L.gdn_spmv_plan_check(self.plan))
        return y

    def close(self):
        self.L.gdn_spmv_plan_free(self.plan)
        for d in (self.d_Ax, self.d_x, self.d_y):
            self.L.gdn_dev_free(d)
        self.L.gdn_graph_free(self.h)


class ResidentBC:
    """Betweenness centrality from many sources on one resident graph (gdn_bc_plan_*: BFS plan + propagation-blocked
    heavy levels)."""

    def __init__(self, g: Graph, with_reverse: bool = True):
        L = _cabi.lib()
        self.L, self.m = L, g.V()
        self.h, self.hi, self.plan = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rp, ci = _arr(g.out_rowptr(), np.uint64), _arr(g.out_colidx(), np.int32)
        _cabi.check(L.gdn_graph_upload(g.V(), g.E(), _p(rp), _p(ci), C.byref(self.h)))
        if with_reverse:
            irp, ici = _arr(g.in_rowptr(), np.uint64), _arr(g.in_colidx(), np.int32)
            _cabi.check(L.gdn_graph_upload(g.V(), g.E(), _p(irp), _p(ici), C.byref(self.hi)))
        _cabi.check(L.gdn_bc_plan_create(self.h, self.hi if with_reverse else None, C.byref(self.plan)))
        self.d_scores = C.c_void_p()
        _cabi.check(L.gdn_dev_alloc(4 * self.m, C.byref(self.d_scores)))

    def run(self, source: int, scores: np.ndarray) -> dict:
        assert scores.dtype == np.float32 and scores.flags.c_contiguous
        _cabi.check(self.L.gdn_dev_upload(self.d_scores, _p(scores), 4 * self.m))
        st = _cabi.GdnStats()
        _cabi.check(self.L.gdn_bc_run(self.plan, int(source), self.d_scores, C.byref(st)))
        _cabi.check(self.L.gdn_dev_download(_p(scores), self.d_scores, 4 * self.m))
        return st.as_dict()

    def close(self):
        self.L.gdn_bc_plan_free(self.plan)
        self.L.gdn_dev_free(self.d_scores)
        if self.hi:
            self.L.gdn_graph_free(self.hi)
        self.L.gdn_graph_free(self.h)


class ResidentPRDelta:
    """Delta PageRank on resident graphs (gdn_pr_delta_plan_*); run() returns (stats, trace)."""

    def __init__(self, g: Graph, layout: int = _cabi.GDN_LAYOUT_AUTO):
        L = _cabi.lib()
        self.L, self.m = L, g.V()
        self.hi, self.ho, self.plan, self.d_scores = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        irp, ici = _arr(g.in_rowptr(), np.uint64), _arr(g.in_colidx(), np.int32)
        orp, oci = _arr(g.out_rowptr(), np.uint64), _arr(g.out_colidx(), np.int32)
        _cabi.check(L.gdn_graph_upload(g.V(), g.E(), _p(irp), _p(ici), C.byref(self.hi)))
        _cabi.check(L.gdn_graph_upload(g.V(), g.E(), _p(orp), _p(oci), C.byref(self.ho)))
        _cabi.check(L.gdn_pr_delta_plan_create(self.hi, self.ho, layout, C.byref(self.plan)))
        _cabi.check(L.gdn_dev_alloc(4 * self.m, C.byref(self.d_scores)))

    def run(self, scores: np.ndarray, damping=0.85, epsilon=1e-4, epsilon2=1e-3, max_iter=100, push_div=8):
        assert scores.dtype == np.float32 and scores.flags.c_contiguous
        _cabi.check(self.L.gdn_dev_upload(self.d_scores, _p(scores), 4 * self.m))
        st = _cabi.GdnStats()
        _cabi.check(self.L.gdn_pr_delta_run(self.plan, self.d_scores, damping, epsilon, epsilon2, max_iter, push_div,
                                            C.byref(st)))
        _cabi.check(self.L.gdn_dev_download(_p(scores), self.d_scores, 4 * self.m))
        n = C.c_int32(0)
        diff, items, mode = np.zeros(max_iter), np.zeros(max_iter, np.int32), np.zeros(max_iter, np.int32)
        _cabi.check(self.L.gdn_pr_delta_trace(self.plan, max_iter, C.byref(n), _p(diff), _p(items), _p(mode)))
        k = n.value
        # mode: 0 pull, 1 push (the reference's print); masked: the push ran as a pull of the frontier's terms
        return st.as_dict(), dict(diff=diff[:k], items=items[:k], mode=mode[:k] & 1, masked=mode[:k] >> 1)

    def close(self):
        self.L.gdn_pr_delta_plan_free(self.plan)
        self.L.gdn_dev_free(self.d_scores)
        self.L.gdn_graph_free(self.ho)
        self.L.gdn_graph_free(self.hi)


class ResidentPageRankShards:
    """The sharded PageRank data path of gardenia_amd.sharded on ONE device: `world` vertex-range
    shards of the same graph, each with its own plan (row_base, m_local < m_global), the all-gather
    emulated by device copies.  Exercises exactly the C-ABI calls a multi-GPU run makes."""

    def __init__(self, g: Graph, world: int, layout: int = _cabi.GDN_LAYOUT_AUTO, parts: int = 1):
        from .sharded import vertex_range
        L = _cabi.lib()
        self.L, self.m, self.world = L, g.V(), world
        self.parts = parts  # > 1: every iteration is issued as row-range parts (gdn_pr_pull_rows_dev)
        irp, ici = _arr(g.in_rowptr(), np.uint64), _arr(g.in_colidx(), np.int32)
        self.h = C.c_void_p()
        _cabi.check(L.gdn_graph_upload(g.V(), g.E(), _p(irp), _p(ici), C.byref(self.h)))
        deg = _arr(g.out_degrees(), np.int32)
        self.ranks = []
        self.chunk = vertex_range(0, world, self.m)[2]
        nfull = self.chunk * world
        for r in range(world):
            lo, hi, _ = vertex_range(r, world, self.m)
            sh, plan = C.c_void_p(), C.c_void_p()
            if hi > lo:
                _cabi.check(L.gdn_graph_slice_rows(self.h, lo, hi, C.byref(sh)))
            d_deg, d_scores, d_diff = C.c_void_p(), C.c_void_p(), C.c_void_p()
            d_c = [C.c_void_p(), C.c_void_p()]
            n = max(hi - lo, 1)
            _cabi.check(L.gdn_dev_alloc(4 * n, C.byref(d_deg)))
            _cabi.check(L.gdn_dev_alloc(4 * n, C.byref(d_scores)))
            _cabi.check(L.gdn_dev_alloc(8, C.byref(d_diff)))
            for k in range(2):
                _cabi.check(L.gdn_dev_alloc(4 * nfull, C.byref(d_c[k])))
                _cabi.check(L.gdn_dev_upload(d_c[k], _p(np.zeros(nfull, np.float32)), 4 * nfull))
            if hi > lo:
                _cabi.check(L.gdn_dev_upload(d_deg, _p(np.ascontiguousarray(deg[lo:hi])), 4 * (hi - lo)))
                init = np.full(hi - lo, np.float32(1.0) / np.float32(self.m), np.float32)
                _cabi.check(L.gdn_dev_upload(d_scores, _p(init), 4 * (hi - lo)))
                _cabi.check(L.gdn_pr_plan_create(sh, d_deg, self.m, lo, layout, C.byref(plan)))
            self.ranks.append(dict(lo=lo, hi=hi, sh=sh, plan=plan, deg=d_deg, scores=d_scores, diff=d_diff, c=d_c))

    def _allgather(self, which):
        # every rank's slice [lo,hi) of buffer `which` -> the same slice of every other rank's buffer
        for src in self.ranks:
            n = src["hi"] - src["lo"]
            if n <= 0:
                continue
            tmp = np.empty(n, np.float32)
            off = 4 * src["lo"]
            _cabi.check(self.L.gdn_dev_download(_p(tmp), C.c_void_p(src["c"][which].value + off), 4 * n))
            for dst in self.ranks:
                if dst is not src:
                    _cabi.check(self.L.gdn_dev_upload(C.c_void_p(dst["c"][which].value + off), _p(tmp), 4 * n))

    def solve(self, epsilon=EPSILON, max_iter=MAX_ITER, damping=0.85):
        L = self.L
        for r in self.ranks:
            if r["plan"]:
                _cabi.check(L.gdn_pr_contrib_dev(r["plan"], r["scores"], r["c"][0], None))
        self._allgather(0)
        cur, it, err = 0, 0, 0.0
        for it in range(max_iter):
            err = 0.0
            for r in self.ranks:
                if r["plan"]:
                    if self.parts <= 1:
                        _cabi.check(L.gdn_pr_pull_dev(r["plan"], r["c"][cur], r["scores"], r["c"][cur ^ 1], r["diff"],
                                                      float(damping), None))
                    else:  # the same part ranges as sharded.ShardedPageRank.part_ranges()
                        from .sharded import ShardedPageRank

                        class _B:
                            pull_rows = None
                        ranges = ShardedPageRank(_B(), self.m, 0, self.world, None, parts=self.parts).part_ranges()
                        n = r["hi"] - r["lo"]
                        for j, (r0, r1) in enumerate(ranges):
                            flags = (_cabi.GDN_PR_PART_FIRST if j == 0 else 0) | \
                                    (_cabi.GDN_PR_PART_LAST if j == len(ranges) - 1 else 0)
                            _cabi.check(L.gdn_pr_pull_rows_dev(r["plan"], r["c"][cur], r["scores"], r["c"][cur ^ 1],
                                                               r["diff"], float(damping), min(r0, n), min(r1, n), flags,
                                                               None))
                    d = np.zeros(1, np.float64)
                    _cabi.check(L.gdn_dev_download(_p(d), r["diff"], 8))
                    err += float(d[0])
            self._allgather(cur ^ 1)
            cur ^= 1
            if err < epsilon:
                break
        scores = np.empty(self.m, np.float32)
        for r in self.ranks:
            if r["plan"]:
                _cabi.check(L.gdn_pr_plan_check(r["plan"]))
                part = np.empty(r["hi"] - r["lo"], np.float32)
                _cabi.check(L.gdn_dev_download(_p(part), r["scores"], 4 * len(part)))
                scores[r["lo"]:r["hi"]] = part
        return scores, it + 1, err

    def close(self):
        for r in self.ranks:
            if r["plan"]:
                self.L.gdn_pr_plan_free(r["plan"])
            if r["sh"]:
                self.L.gdn_graph_free(r["sh"])
            for d in (r["deg"], r["scores"], r["diff"], r["c"][0], r["c"][1]):
                self.L.gdn_dev_free(d)
        self.L.gdn_graph_free(self.h)


class ResidentSSSP:
    """Many SSSP runs on one resident weighted graph (gdn_sssp_plan_*); dense=True adds the
    propagation-blocked Bellman-Ford sweeps for heavy frontiers."""

    def __init__(self, g: Graph, weight: np.ndarray, dense: bool = True):
        L = _cabi.lib()
        self.L, self.m = L, g.V()
        self.h, self.plan, self.d_w, self.d_dist = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        rp, ci = _arr(g.out_rowptr(), np.uint64), _arr(g.out_colidx(), np.int32)
        w = _arr(weight, np.int32)
        _cabi.check(L.gdn_graph_upload(g.V(), g.E(), _p(rp), _p(ci), C.byref(self.h)))
        _cabi.check(L.gdn_dev_alloc(4 * max(g.E(), 1), C.byref(self.d_w)))
        if g.E():
            _cabi.check(L.gdn_dev_upload(self.d_w, _p(w), 4 * g.E()))
        _cabi.check(L.gdn_dev_alloc(4 * g.V(), C.byref(self.d_dist)))
        _cabi.check(L.gdn_sssp_plan_create(self.h, self.d_w, 1 if dense else 0, C.byref(self.plan)))

    def run(self, source: int, delta: int = 1):
        st = _cabi.GdnStats()
        _cabi.check(self.L.gdn_sssp_run(self.plan, source, delta, self.d_dist, C.byref(st)))
        dist = np.empty(self.m, np.int32)
        _cabi.check(self.L.gdn_dev_download(_p(dist), self.d_dist, 4 * self.m))
        return dist, st.as_dict()

    def close(self):
        self.L.gdn_sssp_plan_free(self.plan)
        self.L.gdn_dev_free(self.d_w)
        self.L.gdn_dev_free(self.d_dist)
        self.L.gdn_graph_free(self.h)
