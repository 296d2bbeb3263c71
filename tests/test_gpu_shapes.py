"""Non-R-MAT graph shapes through every solver (tools/shapes.py, small sizes): a 2-D grid (road-like: diameter 500,
degree <= 4), a uniform random graph (no hubs: the record tiers find nothing to take) and a clustered small world
(Watts-Strogatz: many triangles, narrow degrees).  Every thresholded choice in the solvers -- PageRank's tier picker, the
BFS level chooser, SSSP's dense-sweep triggers and fused light phases, TC's formulation, CC's sampling -- was tuned on
R-MAT; here each must still give the oracle's answer."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_every_solver_on_grid_uniform_and_small_world(tmp_path):
    out = str(tmp_path / "shapes.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shapes.py"), "small", out], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=1500, cwd=ROOT,
                       env=dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "16")))
    assert p.returncode == 0, p.stdout[-3000:]
    r = json.load(open(out))
    assert set(r) == {"grid2d", "uniform", "small_world"}
    assert r["grid2d"]["bfs"]["levels"] >= 500 and r["grid2d"]["max_out_degree"] == 4
    assert r["uniform"]["pagerank"]["hubs"] == 0
    assert r["small_world"]["tc"]["triangles"] > r["uniform"]["tc"]["triangles"]
