# Round-3 session 53: long fuzz sweeps with SSSP's record tiers (and BFS heads) forced onto every plan, two weight ranges of the driver
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s53
mkdir -p $O; rm -rf $O/*
export OMP_NUM_THREADS=4
E="FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_BFS_BTD=0 GDN_SSSP_TIER_MIN_NNZ=1 GDN_SSSP_DENSE_IN=100000"
( env $E GDN_SSSP_TIER_MIN_DEG=2 timeout 1300 python3 tests/aids/fuzz_parity.py 1000 1400001 > $O/f_tiers2.txt 2>&1; tail -1 $O/f_tiers2.txt ) &
( env $E GDN_SSSP_TIER_MIN_DEG=1 GDN_SSSP_TIERS=1 timeout 1300 python3 tests/aids/fuzz_parity.py 1000 1500001 > $O/f_tiers1.txt 2>&1; tail -1 $O/f_tiers1.txt ) &
( env $E GDN_SSSP_TIER_MIN_DEG=8 timeout 1300 python3 tests/aids/fuzz_parity.py 1000 1600001 > $O/f_tiers8.txt 2>&1; tail -1 $O/f_tiers8.txt ) &
wait
