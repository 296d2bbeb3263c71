"""The reference's OWN verifiers judge the HIP results: oracle/_ref/ref_* are the reference's verifier.cc files compiled
in place (oracle/Makefile; the prebuilt binaries travel to the GPU box), the graphs reach them through the reference's
own bin loader, and every kernel's GPU output must make them print "Correct"."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import GOLDEN
from gardenia_amd import graphio, solvers

pytestmark = pytest.mark.gpu

REFBIN = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref")
HAVE_REF = all(os.path.exists(os.path.join(REFBIN, b)) for b in ("ref_bfs", "ref_pr", "ref_spmv", "ref_sssp_verify", "ref_cc",
                                                                  "ref_tc", "ref_bc"))


def _run(exe, *args):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    p = subprocess.run([os.path.join(REFBIN, exe), *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       env=env, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:]
    return p.stdout


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("scale,ef,seed", [(12, 16, 11), (15, 12, 12)])
def test_reference_verifiers_accept_the_hip_results(scale, ef, seed):
    g = graphio.rmat_graph(scale, ef, seed=seed)
    gs = graphio.symmetrize(g)
    G = solvers.Graph(csr=g, need_reverse=True)
    Gs = solvers.Graph(csr=gs, in_csr=gs)
    rng = np.random.default_rng(seed)
    s = graphio.first_nonisolated(g)
    with tempfile.TemporaryDirectory() as tmp:
        graphio.write_bin(os.path.join(tmp, "g"), g)
        graphio.write_bin(os.path.join(tmp, "gs"), gs)
        f = lambda name: os.path.join(tmp, name)
        # BFS (src/bfs/verifier.cc)
        dist = np.full(g.m, solvers.MYINFINITY, np.int32)
        solvers.BFSSolver(G, s, dist)
        dist.tofile(f("bfs"))
        assert "Correct" in _run("ref_bfs", "verify", "bin", f("g"), 0, 1, f("bfs"), s)
        # SSSP, weighted (src/sssp/verifier.cc: Dijkstra)
        wt = rng.integers(1, 256, size=g.nnz).astype(np.int32)
        wt.tofile(f("wt"))
        d2 = np.full(g.m, solvers.K_DIST_INF, np.int32)
        solvers.SSSPSolver(G, s, wt, d2, 16)
        d2.tofile(f("sssp"))
        assert "Correct" in _run("ref_sssp_verify", "verify", "bin", f("g"), 0, 0, f("sssp"), s, f("wt"))
        # PageRank (src/pr/verifier.cc: one more push iteration moves less than 1e-4)
        sc = np.full(g.m, np.float32(1.0) / np.float32(g.m), np.float32)
        solvers.PRSolver(G, sc)
        sc.tofile(f("pr"))
        assert "Correct" in _run("ref_pr", "verify", "bin", f("g"), 0, 1, f("pr"))
        # SpMV with the constants of src/spmv/main.cc:29-36 (src/spmv/verifier.cc)
        Ax = np.full(g.nnz, 0.2, np.float32)
        x = np.full(g.m, 0.3, np.float32)
        y = np.zeros(g.m, np.float32)
        solvers.SpmvSolver(G, Ax, x, y)
        y.tofile(f("spmv"))
        assert "Correct" in _run("ref_spmv", "verify", "bin", f("g"), 0, 1, f("spmv"))
        # CC on the symmetrized graph (src/cc/verifier.cc)
        comp = np.arange(gs.m, dtype=np.int32)
        solvers.CCSolver(Gs, comp)
        comp.tofile(f("cc"))
        assert "Correct" in _run("ref_cc", "verify", "bin", f("gs"), 1, 0, f("cc"))
        # BC (src/bc/verifier.cc)
        bc = np.zeros(g.m, np.float32)
        solvers.BCSolver(solvers.Graph(csr=g), s, bc)
        bc.tofile(f("bc"))
        assert "Correct" in _run("ref_bc", "verify", "bin", f("g"), 0, 0, f("bc"), s)
        # TC: the reference recounts on its own bin loader + orientation (src/tc/omp_base.cc, verifier.cc)
        out = _run("ref_tc", f("gs"), f("tc"))
        total = int(np.fromfile(f("tc.total"), dtype=np.uint64)[0])
        assert "Correct" in out
        assert solvers.TCSolver(Gs)[0] == total
