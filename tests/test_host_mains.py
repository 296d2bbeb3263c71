"""The C++ host mirror (gardenia_amd/host): reference-style mains linked against the C-ABI.
CPU: they build, load the fixture graphs with the reference loader's semantics and fail loudly
without a GPU.  GPU: every kernel's main prints "Correct" from its serial verifier."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from gardenia_amd import _cabi, graphio

BIN = os.path.join(ROOT, "gardenia_amd", "host", "bin")
G = os.path.join(GOLDEN, "graphs")


def run(exe, *args):
    p = subprocess.run([os.path.join(BIN, exe), *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    return p.returncode, p.stdout


def test_mains_exist_and_load_graphs_like_the_reference():
    for k in ("bfs", "pr", "pr_delta", "spmv", "sssp", "cc", "tc", "bc"):
        assert os.path.exists(os.path.join(BIN, k + "_hip")), "run __graft_entry__.build()"
    rc, out = run("bfs_hip", "mtx", os.path.join(G, "test_bc"), 1, 0, 0)
    assert "|V| 7 |E| 26" in out  # BASELINE.md known answer
    rc, out = run("cc_hip", "mtx", os.path.join(G, "test_cc"), 1, 0)
    assert "|V| 11 |E| 36" in out and "4 redundent edges are removed" in out
    if _cabi.device_count() == 0:
        assert rc != 0 and "no HIP device" in out  # no CPU fallback


@pytest.mark.gpu
def test_mains_print_correct(tmp_path):
    cases = [("bfs_hip", ["mtx", os.path.join(G, "test_bc"), 0, 1, 0]),
             ("bfs_hip", ["mtx", os.path.join(G, "chesapeake"), 1, 0, 0]),
             ("pr_hip", ["mtx", os.path.join(G, "test_pr"), 0]),
             ("pr_hip", ["mtx", os.path.join(G, "chesapeake"), 1]),
             ("pr_delta_hip", ["mtx", os.path.join(G, "test_pr"), 0]),
             ("pr_delta_hip", ["mtx", os.path.join(G, "chesapeake"), 1]),
             ("spmv_hip", ["mtx", os.path.join(G, "test_bc"), 0, 1]),
             ("sssp_hip", ["mtx", os.path.join(G, "test_bc"), 0, 0, 0, 1]),
             ("cc_hip", ["mtx", os.path.join(G, "test_cc"), 1, 0]),
             ("tc_hip", ["mtx", os.path.join(G, "chesapeake")]),
             ("bc_hip", ["mtx", os.path.join(G, "test_bc"), 1, 0, 0]),
             ("bc_hip", ["mtx", os.path.join(G, "chesapeake"), 1, 0, 0])]
    g = graphio.rmat_graph(14, 16, seed=5)
    graphio.write_bin(str(tmp_path / "rm"), g)
    graphio.write_bin(str(tmp_path / "rms"), graphio.symmetrize(g))
    s = graphio.first_nonisolated(g)
    cases += [("bfs_hip", ["bin", tmp_path / "rm", 0, 1, s]), ("pr_hip", ["bin", tmp_path / "rm", 0]),
              ("pr_delta_hip", ["bin", tmp_path / "rm", 0]),
              ("sssp_hip", ["bin", tmp_path / "rm", 0, 0, s, 2]), ("cc_hip", ["bin", tmp_path / "rms", 1, 0]),
              ("tc_hip", [tmp_path / "rms"]), ("spmv_hip", ["bin", tmp_path / "rm", 0, 1]),
              ("bc_hip", ["bin", tmp_path / "rm", 0, 0, s]), ("bc_hip", ["bin", tmp_path / "rms", 1, 0, s])]
    for exe, args in cases:
        rc, out = run(exe, *args)
        assert rc == 0 and "Correct" in out, (exe, args, out[-600:])
    rc, out = run("pr_hip", "mtx", os.path.join(G, "test_pr"), 0)
    assert "iterations = 15." in out  # test/reference/graph-pr.mtx.out:28
    rc, out = run("tc_hip", "mtx", os.path.join(G, "chesapeake"))
    assert "total_num_triangles = 194" in out
    rc, out = run("pr_delta_hip", "mtx", os.path.join(G, "chesapeake"), 1)
    # the reference's own run of src/pr/omp_delta.cc on this graph: 12 pull iterations, last L1 norm 0.000359
    assert out.count("pull:") == 12 and "pull: 12    0.000359" in out and "iterations = 12." in out
