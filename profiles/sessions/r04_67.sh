# Round-4 session 67: repro of seed 6000914 with the phase trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export OMP_NUM_THREADS=4
GDN_SSSP_TRACE=1 GDN_SSSP_TIER_MIN_NNZ=1 GDN_SSSP_TIER_MIN_DEG=2 GDN_SSSP_DENSE_IN=100000 python3 tools/sssp_fuzz_repro2.py 6000914 2>&1 | tail -40
