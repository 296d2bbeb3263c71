"""The N > 1 paths on DISTINCT devices -- skipped on a 1-GPU box (where tests/test_gpu_multi.py and
test_gpu_bench_sharded.py run the same code with ranks sharing device 0), run wherever a node has two or more GPUs so
that the first real multi-GPU execution is not the driver's scaling run: gdn_pr_multi / gdn_spmv_multi
through RCCL and through peer copies against the single-device bits, the C++ main with GDN_NUM_GPUS=2, and
`bench.py --gpus 2` with one rank per GPU over the RCCL backend of torch.distributed."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from gardenia_amd import _cabi, graphio, solvers

NDEV = _cabi.device_count()
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(NDEV < 2, reason="needs two HIP devices (%d here)" % NDEV)]


def _pr_graph(scale=16, ef=16, seed=47, cut=7):
    g = graphio.rmat_graph(scale, ef, seed=seed)
    m = g.m - cut  # not divisible by the rank count
    src, dst = graphio.csr_to_coo(g)
    keep = (src < m) & (dst < m)
    g = graphio.build_csr(m, src[keep], dst[keep])
    return g, graphio.transpose(g)


@pytest.mark.parametrize("exchange,code", [(None, 1), ("rccl", 1), ("p2p", 2)])
def test_pr_multi_distinct_devices_bits_equal_single_device(orc, monkeypatch, exchange, code):
    monkeypatch.setenv("GDN_PR_LAYOUT", "pb")
    monkeypatch.setenv("GDN_PB_HUB_MIN_NNZ", "1")
    if exchange:
        monkeypatch.setenv("GDN_MULTI_EXCHANGE", exchange)
    g, gi = _pr_graph()
    G = solvers.Graph(csr=g, in_csr=gi)
    one = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    st1 = solvers.PRSolver(G, one)
    for devs in ([0, 1], list(range(min(NDEV, 8)))):
        many = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        stn = solvers.PRSolver(G, many, devices=devs)
        assert stn["reserved"] == code, (devs, stn)  # 1: the RCCL all-gather ran, 2: peer copies
        assert stn["iterations"] == st1["iterations"]
        assert np.array_equal(one.view(np.uint32), many.view(np.uint32)), devs
        np.testing.assert_allclose(stn["trace"], st1["trace"], rtol=1e-9)
    want, it, _ = orc.pr(gi, g.degrees())
    assert it == st1["iterations"]
    np.testing.assert_allclose(one, want, rtol=1e-4, atol=0)


def test_pr_multi_max_iter_limited_reports_like_gdn_pr():
    """A solve cut off by max_iter reports MAX_ITER + 1 iterations on one device and on two (src/pr/omp_base.cc:39)."""
    g, gi = _pr_graph(14, 16, 48, 3)
    G = solvers.Graph(csr=g, in_csr=gi)
    a = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
    b = a.copy()
    st1 = solvers.PRSolver(G, a, max_iter=3)
    st2 = solvers.PRSolver(G, b, max_iter=3, devices=[0, 1])
    assert st1["iterations"] == st2["iterations"] == 4
    np.testing.assert_allclose(a, b, rtol=1e-6)


def test_spmv_multi_distinct_devices(orc):
    g = graphio.rmat_graph(15, 16, seed=49)
    gi = graphio.transpose(g)
    G = solvers.Graph(csr=g, in_csr=gi)
    rng = np.random.default_rng(11)
    Ax, x, y0 = (rng.random(n).astype(np.float32) for n in (g.nnz, g.m, g.m))
    y1, yn = y0.copy(), y0.copy()
    solvers.SpmvSolver(G, Ax, x, y1)
    solvers.SpmvSolver(G, Ax, x, yn, devices=[0, 1])
    assert np.array_equal(y1.view(np.uint32), yn.view(np.uint32))  # merge-path row sums do not depend on the cut
    want = orc.spmv(gi, Ax, x, y0)
    assert orc.spmv_max_rel_error(yn, want) <= 5 * np.sqrt(np.finfo(np.float32).eps)


def test_pr_main_on_two_gpus():
    exe = os.path.join(ROOT, "gardenia_amd", "host", "bin", "pr_hip")
    p = subprocess.run([exe, "mtx", os.path.join(GOLDEN, "graphs", "chesapeake"), "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300, env=dict(os.environ, GDN_NUM_GPUS="2"))
    assert p.returncode == 0 and "Correct" in p.stdout and "2 GPUs" in p.stdout, p.stdout[-800:]


def test_bench_two_gpus_one_rank_each_over_rccl():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "22", "--steps", "4", "--warmup", "1", "--no-bfs",
            "--no-cpu", "--no-extras"]
    lines = {}
    for n in (1, 2):
        out = subprocess.run(base + ["--gpus", str(n)], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        lines[n] = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert lines[2]["n_gpus"] == 2 and "RCCL all-gather" in lines[2]["config"]["partition"]
    one, two = lines[1]["pr_last_l1_change"], lines[2]["pr_last_l1_change"]
    assert abs(one - two) <= 1e-12 * one  # integer row sums: the same state whatever the cut
