# Round-4 session 110: randomised sweep on the final code: default, every tier forced (+ the TC core), old builder, default under the allocation fence
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s110
mkdir -p $O; rm -rf $O/*
export OMP_NUM_THREADS=4
T="GDN_TC_FORM=f GDN_TC_CORE=4096 GDN_PR_LAYOUT=p GDN_SPMV_LAYOUT=p GDN_PRD_LAYOUT=p GDN_PB_HUB_MIN_NNZ=1 GDN_PB_HUB_MIN=8 GDN_PB_MID_CAP=300 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_SSSP_TIER_MIN_NNZ=1 GDN_SSSP_TIER_MIN_DEG=2"
rep() { echo "$1: $(grep -B4 'Memory access fault' $O/$1.txt | head -5 | tr '\n' ' ' | cut -c1-300) $(tail -1 $O/$1.txt | cut -c1-100)"; }
( FUZZ_PLANS=1 timeout 1500 python3 tests/aids/fuzz_parity.py 400 61000001 > $O/default.txt 2>&1; rep default ) &
( env FUZZ_PLANS=1 $T timeout 1500 python3 tests/aids/fuzz_parity.py 400 62000001 > $O/tiers.txt 2>&1; rep tiers ) &
( env FUZZ_PLANS=1 $T GDN_PB_BUILDER=old GDN_BFS_BTD=0 timeout 1500 python3 tests/aids/fuzz_parity.py 400 63000001 > $O/old.txt 2>&1; rep old ) &
( FUZZ_PLANS=1 GDN_ALLOC_FENCE=1 timeout 1500 python3 tests/aids/fuzz_parity.py 300 64000001 > $O/fence.txt 2>&1; rep fence ) &
( env FUZZ_PLANS=1 $T GDN_ALLOC_FENCE=1 timeout 1500 python3 tests/aids/fuzz_parity.py 300 65000001 > $O/tiers_fence.txt 2>&1; rep tiers_fence ) &
wait
