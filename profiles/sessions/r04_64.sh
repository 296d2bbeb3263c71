# Round-4 session 64: the one mismatch of the extended sweep (seed 6000609, PageRank shards): same seed under the knobs of that mode and without
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export OMP_NUM_THREADS=4
FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_BFS_BTD=0 GDN_SSSP_TIER_MIN_NNZ=1 GDN_SSSP_TIER_MIN_DEG=2 GDN_SSSP_DENSE_IN=100000 python3 tests/aids/fuzz_parity.py 1 6000609 2>&1 | tail -2
FUZZ_PLANS=1 python3 tests/aids/fuzz_parity.py 1 6000609 2>&1 | tail -2
FUZZ_PLANS=1 GDN_PB_V_IL=0 GDN_PB_REC_IL=0 python3 tests/aids/fuzz_parity.py 1 6000609 2>&1 | tail -2
FUZZ_PLANS=1 GDN_PB_BUILDER=old python3 tests/aids/fuzz_parity.py 1 6000609 2>&1 | tail -2
