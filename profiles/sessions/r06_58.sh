# Round-6 session 58: 3 x 400 fresh graphs of the randomised sweep through the BFS plans with head records (bottom-up forced on every heavy level: the flat scan in rounds), ranks for every head, deferred depths; and the TC forward count with its core on 40 seeded R-MAT graphs of scale 12-16 against the oracle
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s58
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1 OMP_NUM_THREADS=4
( FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_BFS_BTD=0 timeout 1500 python3 tests/aids/fuzz_parity.py 400 61000001 > $O/fuzz_heads.txt 2>&1; echo "heads: $(tail -1 $O/fuzz_heads.txt | cut -c1-120)" ) &
( FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_BFS_BTD=0 GDN_BFS_HUBS2=1 GDN_BFS_REC_COMPACT=1 GDN_BFS_DEFER_DEPTH=1 timeout 1500 python3 tests/aids/fuzz_parity.py 400 62000001 > $O/fuzz_heads2.txt 2>&1; echo "heads2: $(tail -1 $O/fuzz_heads2.txt | cut -c1-120)" ) &
( FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_DEFER_DEPTH=1 GDN_BFS_REC_COMPACT=1 GDN_BFS_TD_DEFER_MIN=1 GDN_BFS_SNAP_MIN=1 GDN_BFS_TD_BLIND=1 timeout 1500 python3 tests/aids/fuzz_parity.py 400 63000001 > $O/fuzz_blind.txt 2>&1; echo "deferred+blind: $(tail -1 $O/fuzz_blind.txt | cut -c1-120)" ) &
wait
timeout 1200 python3 - > $O/tc_sweep.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from gardenia_amd import graphio, solvers
from oracle import binding as orc
bad = 0
n = 0
for seed in range(40):
    scale = 12 + seed % 5
    g = graphio.symmetrize(graphio.rmat_graph(scale, 8 + (seed % 3) * 8, seed=1000 + seed))
    want = orc.tc(orc.tc_orient(g))
    for form, core in (("f", "4096"), ("f", "0"), ("a", None)):
        os.environ["GDN_TC_FORM"] = form
        if core is None: os.environ.pop("GDN_TC_CORE", None)
        else: os.environ["GDN_TC_CORE"] = core
        total, st = solvers.TCSolver(solvers.Graph(csr=g, symmetrize=True))
        n += 1
        if total != want:
            bad += 1
            print("MISMATCH", seed, scale, form, core, total, want)
print("tc sweep: %d counts, %d mismatches" % (n, bad))
PY
tail -2 $O/tc_sweep.txt
