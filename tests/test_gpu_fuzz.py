"""Randomised parity sweep (tests/aids/fuzz_parity.py): random graphs of six shapes through every drop-in solver of the
C-ABI, each result against the CPU oracle.  Run as a child process with four OpenMP threads for the oracle: on the GPU
box's 128 hardware threads an OpenMP region per tiny graph costs seconds."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first_seed,blocked", [(1, False), (100001, False), (200001, True), (300001, "plans"),
                                                (700001, "fused"), (800001, "heads"), (900001, "heads2"), (1000001, "deferred")])
def test_random_graphs_every_solver_equals_the_oracle(first_seed, blocked):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    if blocked == "fused":  # every BFS level, every forward and every backward level of BC that fits, and PageRank's
        # whole solve inside the one-workgroup kernels
        env.update(GDN_BFS_SMALL_NF="100000", GDN_BFS_SMALL_SCOUT="1000000000", GDN_BC_SMALL_NF="1000000",
                   GDN_BC_SMALL_SCOUT="1000000000000", GDN_BC_BACK_NF="1024", GDN_BC_BACK_SCOUT="1000000000000",
                   GDN_PR_FUSED="1", GDN_PR_SMALL_M="16384", FUZZ_PLANS="1")
        blocked = False
    if blocked == "deferred":  # BFS plans with head records whose heavy levels -- binned top-down forced below a third of the edges,
        # dense sweeps, bottom-up steps -- all leave the distances to the pass at the end of the search (default from 2^25 vertices on)
        env.update(FUZZ_PLANS="1", GDN_BFS_HEADS_MIN_NNZ="1", GDN_BFS_HUB_MIN="0", GDN_BFS_DEFER_DEPTH="1", GDN_BFS_REC_COMPACT="1", GDN_BFS_TD_DEFER_MIN="1", GDN_BFS_SNAP_MIN="1",
                   GDN_BFS_BTD="2", GDN_BFS_ALPHA_BTD="100000", GDN_BFS_BTD_MIN="1")
        blocked = False
    if blocked == "heads2":  # ... and every head named by rank (the outer hubs of BFS's bottom-up step), the records read from their compact copy
        env.update(GDN_BFS_HUBS2="1", GDN_BFS_REC_COMPACT="1", GDN_BFS_DEFER_DEPTH="1")
        blocked = "heads"
    if blocked == "heads":  # the resident plans with every heavy BFS level on the bottom-up step and its head records
        # (normally from 2^24 edges on; hub test always on), and SSSP's sweeps with their record tiers (from 2^22 edges on)
        env.update(FUZZ_PLANS="1", GDN_BFS_HEADS_MIN_NNZ="1", GDN_BFS_HUB_MIN="0", GDN_BFS_BU_EDGE_DIV="1000000000",
                   GDN_BFS_BTD="0", GDN_SSSP_TIER_MIN_NNZ="1", GDN_SSSP_TIER_MIN_DEG="2", GDN_SSSP_DENSE_IN="100000")
        blocked = False
    if blocked == "plans":  # also the resident plans: dense BFS / SSSP sweeps, BC's blocked levels -- with the binned
        # top-down level forced onto every heavy BFS level below a third of the edges
        env.update(FUZZ_PLANS="1", GDN_BFS_BTD="2", GDN_BFS_ALPHA_BTD="100000", GDN_BFS_BTD_MIN="1")
        blocked = False
    if blocked:  # the propagation-blocked layouts with their record tiers on these small graphs too (normally >= 2^22 edges)
        env.update(GDN_PR_LAYOUT="p", GDN_SPMV_LAYOUT="p", GDN_PRD_LAYOUT="p", GDN_PB_HUB_MIN_NNZ="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "aids", "fuzz_parity.py"), "200", str(first_seed)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and "every solver equal to the oracle" in p.stdout, p.stdout[-2000:]
