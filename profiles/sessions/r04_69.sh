# Round-4 session 69: more of the randomised sweep on the final code: 2500 fresh graphs in each of five modes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s69
mkdir -p $O; rm -rf $O/*
export OMP_NUM_THREADS=4
( timeout 3000 python3 tests/aids/fuzz_parity.py 2500 11000001 > $O/plain.txt 2>&1; grep -E "borderline|MISMATCH|fuzz parity" $O/plain.txt ) &
( GDN_PR_LAYOUT=p GDN_SPMV_LAYOUT=p GDN_PRD_LAYOUT=p GDN_PB_HUB_MIN_NNZ=1 timeout 3000 python3 tests/aids/fuzz_parity.py 2500 12000001 > $O/blocked.txt 2>&1; grep -E "borderline|MISMATCH|fuzz parity" $O/blocked.txt ) &
( FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_BFS_BTD=0 GDN_SSSP_TIER_MIN_NNZ=1 GDN_SSSP_TIER_MIN_DEG=2 GDN_SSSP_DENSE_IN=100000 timeout 3000 python3 tests/aids/fuzz_parity.py 2500 13000001 > $O/heads.txt 2>&1; grep -E "borderline|MISMATCH|fuzz parity" $O/heads.txt ) &
( FUZZ_PLANS=1 GDN_BFS_BTD=2 GDN_BFS_ALPHA_BTD=100000 GDN_BFS_BTD_MIN=1 timeout 3000 python3 tests/aids/fuzz_parity.py 2500 14000001 > $O/plans.txt 2>&1; grep -E "borderline|MISMATCH|fuzz parity" $O/plans.txt ) &
wait
( GDN_BFS_SMALL_NF=100000 GDN_BFS_SMALL_SCOUT=1000000000 GDN_BC_SMALL_NF=1000000 GDN_BC_SMALL_SCOUT=1000000000000 GDN_BC_BACK_NF=1024 GDN_BC_BACK_SCOUT=1000000000000 GDN_PR_FUSED=1 GDN_PR_SMALL_M=16384 FUZZ_PLANS=1 timeout 3000 python3 tests/aids/fuzz_parity.py 2500 15000001 > $O/fused.txt 2>&1; grep -E "borderline|MISMATCH|fuzz parity" $O/fused.txt ) &
( FUZZ_PLANS=1 GDN_PR_LAYOUT=p GDN_SPMV_LAYOUT=p GDN_PRD_LAYOUT=p GDN_PB_HUB_MIN_NNZ=1 GDN_PB_HUB_MIN=8 GDN_PB_MID_CAP=300 GDN_SSSP_TIER_MIN_NNZ=1 GDN_SSSP_TIER_MIN_DEG=2 timeout 3000 python3 tests/aids/fuzz_parity.py 2500 16000001 > $O/tiers.txt 2>&1; grep -E "borderline|MISMATCH|fuzz parity" $O/tiers.txt ) &
wait
