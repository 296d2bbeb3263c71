# Round-6 session 44: BFS on a uniform random graph (2^26 x 16) with the flat bottom-up scan: engine choice of the heavy levels again
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s44
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1 GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_exp/libgardenia_hip.so
timeout 1500 python3 tools/bfs_shapes_trace.py uniform26 > $O/uniform26.txt 2>&1; grep -E "^uniform26|level [678]" $O/uniform26.txt | head -60
