# Round-4 session 63: extended randomised parity sweep on the final code: 1000 graphs per mode (fresh seeds), the modes of tests/test_gpu_fuzz.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s63
mkdir -p $O; rm -rf $O/*
export OMP_NUM_THREADS=4
( timeout 1500 python3 tests/aids/fuzz_parity.py 1000 4000001 > $O/plain.txt 2>&1; tail -1 $O/plain.txt ) &
( GDN_PR_LAYOUT=p GDN_SPMV_LAYOUT=p GDN_PRD_LAYOUT=p GDN_PB_HUB_MIN_NNZ=1 timeout 1500 python3 tests/aids/fuzz_parity.py 1000 5000001 > $O/blocked.txt 2>&1; tail -1 $O/blocked.txt ) &
wait
( FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_BFS_BTD=0 GDN_SSSP_TIER_MIN_NNZ=1 GDN_SSSP_TIER_MIN_DEG=2 GDN_SSSP_DENSE_IN=100000 timeout 1500 python3 tests/aids/fuzz_parity.py 1000 6000001 > $O/heads.txt 2>&1; tail -1 $O/heads.txt ) &
( FUZZ_PLANS=1 GDN_BFS_BTD=2 GDN_BFS_ALPHA_BTD=100000 GDN_BFS_BTD_MIN=1 timeout 1500 python3 tests/aids/fuzz_parity.py 1000 7000001 > $O/plans.txt 2>&1; tail -1 $O/plans.txt ) &
wait
