# Round-4 session 66: the SSSP-plan mismatch of the extended sweep (seed 6000914): which of the round's changes it follows
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export OMP_NUM_THREADS=4
B="FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_BFS_BTD=0 GDN_SSSP_TIER_MIN_NNZ=1 GDN_SSSP_TIER_MIN_DEG=2 GDN_SSSP_DENSE_IN=100000"
for extra in "" "GDN_SSSP_REC_IL=0" "GDN_SSSP_ADAPT=0" "GDN_SSSP_SMALL_FAR=65536" "GDN_PB_BUILDER=old" "GDN_SSSP_TIERS=0"; do
  echo "== $extra"; env $B $extra python3 tests/aids/fuzz_parity.py 1 6000914 2>&1 | tail -1
done
