# Round-6 session 48: the closing pass of a BFS search instantiated per number of kept levels: tests, the three sources, its kernel time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s48
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "bfs" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python3 tools/bfs_runs.py 27 4 1 > $O/bfs27.txt 2>&1; grep -E "^round|distances of" $O/bfs27.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bfs_runs.py 27 3 1 > $O/bfs.txt 2>&1
grep -h "bfs_depth_finish" $O/trace/*/*_kernel_stats.csv | cut -d'"' -f2,3 | cut -c1-60,200-
for s in 24 26; do timeout 600 python3 tools/traffic_run.py bfs $s 6 2>&1 | tail -1; done
