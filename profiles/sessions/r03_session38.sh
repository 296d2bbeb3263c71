# Round-3 session 38: longer fuzz sweeps over the code changed late in the round (heads, per-workgroup flushes, CC link election, mailbox)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s38
mkdir -p $O; rm -rf $O/*
export OMP_NUM_THREADS=4
( timeout 1100 python3 tests/aids/fuzz_parity.py 600 900001 > $O/fuzz_default.txt 2>&1; tail -1 $O/fuzz_default.txt ) &
( env FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_BFS_BTD=0 timeout 1100 python3 tests/aids/fuzz_parity.py 600 910001 > $O/fuzz_heads.txt 2>&1; tail -1 $O/fuzz_heads.txt ) &
( env FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 timeout 1100 python3 tests/aids/fuzz_parity.py 600 920001 > $O/fuzz_plans.txt 2>&1; tail -1 $O/fuzz_plans.txt ) &
wait
