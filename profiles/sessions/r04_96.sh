# Round-4 session 96: after the TC core: TC tests of every file, the full-size TC config, sweeps under the allocation fence with the core forced
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s96
mkdir -p $O; rm -rf $O/*
export OMP_NUM_THREADS=4
T="FUZZ_PLANS=1 GDN_TC_FORM=f GDN_TC_CORE=4096 GDN_PR_LAYOUT=p GDN_SPMV_LAYOUT=p GDN_PRD_LAYOUT=p GDN_PB_HUB_MIN_NNZ=1 GDN_PB_HUB_MIN=8 GDN_PB_MID_CAP=300 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_SSSP_TIER_MIN_NNZ=1 GDN_SSSP_TIER_MIN_DEG=2"
rep() { echo "$1: $(grep -B4 'Memory access fault' $O/$1.txt | head -5 | tr '\n' ' ' | cut -c1-300) $(tail -1 $O/$1.txt | cut -c1-100)"; }
( env $T GDN_ALLOC_FENCE=1 timeout 2400 python3 tests/aids/fuzz_parity.py 300 41000001 > $O/tiers_fence.txt 2>&1; rep tiers_fence ) &
( env $T timeout 2400 python3 tests/aids/fuzz_parity.py 500 42000001 > $O/tiers.txt 2>&1; rep tiers ) &
( GDN_ALLOC_FENCE=1 timeout 2400 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider -k "tc or TC" > $O/tc_fence.txt 2>&1; echo "tc_fence: $(grep -i 'passed\|failed' $O/tc_fence.txt | tail -1)" ) &
wait
timeout 2400 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/suite.txt 2>&1; echo "suite: $(grep -i 'passed\|failed' $O/suite.txt | tail -1)"
