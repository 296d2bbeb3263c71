# Round-6 session 45: the flat scan in rounds (quota 4, 8, 16, ... in-edges per open row): BFS tests, RMAT-27 sources, the uniform graph with its heavy level forced bottom-up
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s45
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "bfs" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python3 tools/bfs_runs.py 27 4 1 > $O/bfs27.txt 2>&1; grep -E "^round|level [345] bottom" $O/bfs27.txt
for s in 24 22 26; do timeout 600 python3 tools/traffic_run.py bfs $s 6 2>&1 | tail -1; done
export GDN_TEST_HOOKS=1 GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_exp/libgardenia_hip.so
timeout 1500 python3 tools/bfs_shapes_trace.py uniform26 > $O/uniform26.txt 2>&1; grep -E "^uniform26|level [678]" $O/uniform26.txt | head -40
