# Round-6 session 38: how long a late level stays on the bottom-up engine (GDN_BFS_BU_STAY: while the frontier scouts more than m / stay edges), on the flat scan
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s38
mkdir -p $O; rm -rf $O/*
export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_exp/libgardenia_hip.so
for k in 256 64 32 16 8; do GDN_BFS_BU_STAY=$k timeout 600 python3 tools/bfs_runs.py 27 4 1 > $O/bfs_stay$k.txt 2>&1; echo "== GDN_BFS_BU_STAY=$k"; grep -E "^round" $O/bfs_stay$k.txt; done
grep -E "level|traced" $O/bfs_stay16.txt | head -40
