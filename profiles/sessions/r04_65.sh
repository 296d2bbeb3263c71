# Round-4 session 65: the rest of the extended sweep in the "heads" mode (seeds 6000609 .. 6001000) + 600 more graphs in the "fused" mode
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s65
mkdir -p $O; rm -rf $O/*
export OMP_NUM_THREADS=4
( FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_BFS_BTD=0 GDN_SSSP_TIER_MIN_NNZ=1 GDN_SSSP_TIER_MIN_DEG=2 GDN_SSSP_DENSE_IN=100000 timeout 1500 python3 tests/aids/fuzz_parity.py 392 6000609 > $O/heads.txt 2>&1; grep -E "borderline|MISMATCH|fuzz parity" $O/heads.txt ) &
( GDN_BFS_SMALL_NF=100000 GDN_BFS_SMALL_SCOUT=1000000000 GDN_BC_SMALL_NF=1000000 GDN_BC_SMALL_SCOUT=1000000000000 GDN_BC_BACK_NF=1024 GDN_BC_BACK_SCOUT=1000000000000 GDN_PR_FUSED=1 GDN_PR_SMALL_M=16384 FUZZ_PLANS=1 timeout 1500 python3 tests/aids/fuzz_parity.py 600 8000001 > $O/fused.txt 2>&1; grep -E "borderline|MISMATCH|fuzz parity" $O/fused.txt ) &
wait
