"""N>1 path on CPU: world_size-2 (and 3) gloo jobs run the product's sharded PageRank
orchestration (partition, in-place all-gather of the contrib vector, all-reduce of the L1
change, convergence test) with a numpy stand-in for the HIP kernel, and must reproduce the
single-process oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from gardenia_amd import graphio
from gardenia_amd.sharded import vertex_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_vertex_range_partition():
    for m, w in [(10, 2), (10, 3), (7, 8), (1 << 20, 8)]:
        ranges = [vertex_range(r, w, m) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == m
        for a, b in zip(ranges, ranges[1:]):
            assert a[1] == b[0]
        assert all(hi - lo <= c for lo, hi, c in ranges) and ranges[0][2] * w >= m


def test_part_ranges_cover_the_chunk_and_agree_across_ranks():
    from gardenia_amd.sharded import ShardedPageRank

    class B:
        def pull_rows(self):
            pass
    for m, world, parts in [(1000, 2, 4), (1 << 20, 8, 4), (37, 3, 4), (64, 8, 3), (5, 2, 4)]:
        got = [ShardedPageRank(B(), m, r, world, dist=None, parts=parts).part_ranges() for r in range(world)]
        assert all(g == got[0] for g in got)
        chunk = vertex_range(0, world, m)[2]
        assert got[0][0][0] == 0 and got[0][-1][1] == chunk
        for (a0, a1), (b0, b1) in zip(got[0], got[0][1:]):
            assert a1 == b0 and a0 % 4 == 0
    assert ShardedPageRank(B(), 100, 0, 1, dist=None, parts=4).parts == 1  # a single rank is never cut


@pytest.mark.parametrize("world,parts,exchange,balanced,tickets",
                         [(2, 4, "dense", 0, 0), (3, 4, "dense", 0, 0), (2, 1, "dense", 0, 0), (3, 7, "dense", 0, 0),
                          (2, 4, "compact", 0, 0), (3, 3, "compact", 0, 0), (2, 1, "compact", 0, 0),
                          (2, 4, "dense", 1, 0), (3, 3, "compact", 1, 0), (3, 1, "dense", 1, 0),
                          (2, 4, "dense", 0, 1), (3, 3, "compact", 1, 1), (3, 5, "dense", 1, 1)])
def test_sharded_pagerank_gloo(orc, tmp_path, world, parts, exchange, balanced, tickets):
    """balanced = 1: nnz-balanced vertex ranges in the padded vertex space, as bench.py --gpus N cuts them.
    tickets = 1 (round 6): the backend computes the whole iteration at once and releases a part's rows only inside
    part_ready(j) -- the driver's contract with gdn_pr_pull_parts_dev / gdn_pr_wait_part_dev: every exchange is queued inside
    its part's context, after the part was released (sharded_worker.TicketedNumpyBackend poisons what is not released)."""
    scale, ef = 8, 8
    out = str(tmp_path / "pr")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", GDN_TEST_PARTS=str(parts), GDN_TEST_EXCHANGE=exchange,
                   GDN_TEST_BALANCED=str(balanced), GDN_TEST_TICKETS=str(tickets))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"),
                                       str(scale), str(ef), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    # single-process oracle on the same graph
    g = graphio.rmat_graph(scale, ef, seed=77)
    m = g.m - 3
    src, dst = graphio.csr_to_coo(g)
    keep = (src < m) & (dst < m)
    g = graphio.build_csr(m, src[keep], dst[keep])
    gi = graphio.transpose(g)
    want, it, trace = orc.pr(gi, g.degrees())
    got = np.concatenate([np.load(f"{out}.{r}.npy") for r in range(world)])
    if balanced:  # the cut follows the edges, not the vertex count
        b = np.load(f"{out}.bounds.npy")
        per = np.diff(gi.rowptr[b].astype(np.int64))
        assert per.max() <= gi.nnz / world + gi.degrees().max() + 1
    meta = np.load(f"{out}.meta.npy")
    assert int(meta[0]) == it
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=0)
    assert abs(meta[1] - trace[-1]) < 1e-9


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_spmv_gloo(orc, tmp_path, world):
    """ShardedSpMV: all-gather of x, local multiply of the row range, y distributed."""
    out = str(tmp_path / "spmv")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_spmv_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    g = graphio.rmat_graph(8, 8, seed=78)
    rng = np.random.default_rng(5)
    Ax = rng.random(g.nnz).astype(np.float32)
    x = rng.random(g.m).astype(np.float32)
    y0 = rng.random(g.m).astype(np.float32)
    want = orc.spmv(g, Ax, x, y0)
    got = np.concatenate([np.load(f"{out}.{r}.npy") for r in range(world)])
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=0)


def test_edge_balanced_ranges():
    from gardenia_amd.sharded import edge_balanced_ranges
    rp = np.array([0, 0, 10, 10, 11, 30, 30, 31, 40], np.uint64)
    for w in (1, 2, 3, 8, 11):
        r = edge_balanced_ranges(rp, w)
        assert len(r) == w and r[0][0] == 0 and r[-1][1] == 8
        assert all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(r, r[1:]))
    assert edge_balanced_ranges(rp, 2) == [(0, 5), (5, 8)]  # the first row boundary at or behind nnz / 2 edges
    assert edge_balanced_ranges(np.zeros(4, np.uint64), 3)[-1] == (0, 3)  # no edges: everything to the last rank
    for w in (1, 2, 3, 8):  # min_rows = 1: nobody is left without a row
        r = edge_balanced_ranges(rp, w, min_rows=1)
        assert r[0][0] == 0 and r[-1][1] == 8 and all(hi > lo for lo, hi in r)
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert edge_balanced_ranges(np.zeros(4, np.uint64), 3, min_rows=1) == [(0, 1), (1, 2), (2, 3)]


def test_pad_columns_and_padded_chunk():
    from gardenia_amd.sharded import pad_columns, padded_chunk
    bounds = [0, 5, 6, 13]
    chunk = padded_chunk(bounds)
    assert chunk == 8
    got = pad_columns(np.array([0, 4, 5, 6, 12], np.int32), bounds, chunk)
    assert got.tolist() == [0, 4, 8, 16, 22]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_tc_gloo(orc, tmp_path, world):
    """Row ranges of equal DAG-edge count, partial counts, one all-reduce."""
    scale, ef = 7, 8
    out = str(tmp_path / "tc.npy")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_tc_worker.py"),
                                       str(scale), str(ef), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    g = graphio.symmetrize(graphio.rmat_graph(scale, ef, seed=78))
    dag = graphio.orient_dag(g)
    want = orc.tc(dag)
    got = np.load(out)
    assert int(got[0]) == want > 0
    assert 0 == got[1] < got[2] < dag.m  # rank 0 counted a proper part
