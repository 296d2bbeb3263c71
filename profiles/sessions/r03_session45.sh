# Round-3 session 45: long fuzz sweeps on the final code (default / blocked layouts / plans with heads / fused kernels)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s45
mkdir -p $O; rm -rf $O/*
export OMP_NUM_THREADS=4
( timeout 1300 python3 tests/aids/fuzz_parity.py 1200 1000001 > $O/f_default.txt 2>&1; tail -1 $O/f_default.txt ) &
( env GDN_PR_LAYOUT=p GDN_SPMV_LAYOUT=p GDN_PRD_LAYOUT=p GDN_PB_HUB_MIN_NNZ=1 GDN_PLACE_MIN_EDGES=1 GDN_PLACE_MIN_BYTES=1 GDN_PR_PLACE=1 GDN_SPMV_PLACE=1 timeout 1300 python3 tests/aids/fuzz_parity.py 800 1100001 > $O/f_blocked_place.txt 2>&1; tail -1 $O/f_blocked_place.txt ) &
( env FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_BFS_BTD=0 timeout 1300 python3 tests/aids/fuzz_parity.py 1000 1200001 > $O/f_heads.txt 2>&1; tail -1 $O/f_heads.txt ) &
( env FUZZ_PLANS=1 GDN_BFS_BTD=2 GDN_BFS_ALPHA_BTD=100000 GDN_BFS_BTD_MIN=1 GDN_MAILBOX=0 timeout 1300 python3 tests/aids/fuzz_parity.py 1000 1300001 > $O/f_plans_nomail.txt 2>&1; tail -1 $O/f_plans_nomail.txt ) &
wait
