# Round-6 session 18: mid-size PageRank, one round of large chunks against two rounds of half-size ones (experiments build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s18
mkdir -p $O; rm -rf $O/*
export GARDENIA_HIP_LIB=$GRAFT_REPO_ROOT/gardenia_amd/lib/var_exp/libgardenia_hip.so GDN_TEST_HOOKS=1
S='"" "GDN_PB_LOG_CHUNK=15,GDN_PB_FILL_ROUND=1" "GDN_PB_LOG_CHUNK=15" "GDN_PB_SLICES_LOG=8,GDN_PB_FILL_ROUND=1" "GDN_PB_LOG_CHUNK=13" "GDN_PB_BALANCE_ALL=1,GDN_PB_LOG_CHUNK=15,GDN_PB_FILL_ROUND=1"'
eval timeout 900 python3 tools/pr_midsize.py $S > $O/lj.txt 2>&1; tail -7 $O/lj.txt
for sc in 22 24 25; do
  eval PR_MIDSIZE_RMAT=$sc timeout 900 python3 tools/pr_midsize.py $S > $O/rmat$sc.txt 2>&1; tail -7 $O/rmat$sc.txt
done
