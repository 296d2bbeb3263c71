# Round-6 session 35: the bottom-up step's second stage as a flat scan (all in-edges of 64 open rows as one list): BFS tests, the three bench sources, old forms behind the knob
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s35
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "bfs" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python3 tools/bfs_runs.py 27 4 1 > $O/bfs_flat.txt 2>&1; grep -E "^round|level [345] bottom|traced" $O/bfs_flat.txt
export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_exp/libgardenia_hip.so
for k in 1 4; do GDN_BFS_BU_SCAN=$k timeout 600 python3 tools/bfs_runs.py 27 4 1 > $O/bfs_scan$k.txt 2>&1; echo "== GDN_BFS_BU_SCAN=$k"; grep -E "^round|level 3 bottom" $O/bfs_scan$k.txt; done
unset GARDENIA_HIP_LIB
for s in 24 22 26; do timeout 600 python3 tools/traffic_run.py bfs $s 6 2>&1 | tail -1; done
