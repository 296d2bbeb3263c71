# Round-3 session 39: the binned top-down level (source 5 of RMAT-27; RMAT-28): per-level trace + tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s39
mkdir -p $O; rm -rf $O/*
timeout 600 python3 -m pytest tests -m gpu -q -x -k "binned or bfs_resident or fuzz" > $O/pytest.txt 2>&1; grep "passed\|failed" $O/pytest.txt | tail -1
env GDN_BFS_TRACE=1 timeout 300 python3 tools/bfs_notorch.py 27 2>&1 | grep "binned\|BFS RMAT" > $O/b27.txt; cat $O/b27.txt
timeout 300 python3 tools/bfs_notorch.py 28 2>&1 | grep "BFS RMAT" > $O/b28.txt; cat $O/b28.txt
timeout 600 python3 -m pytest tests -m gpu -q -x -k "sssp or SSSP" > $O/pytest2.txt 2>&1; grep "passed\|failed" $O/pytest2.txt | tail -1
for k in "16 rand" "1 unit"; do REPS=6 timeout 300 python3 tools/sssp_trace.py 24 $k plan 2>&1 | grep "RMAT" | awk '{print $3, $4, $6}' | tr '\n' ' '; echo; done
REPS=4 timeout 300 python3 tools/sssp_trace.py 26 16 rand plan 2>&1 | grep "RMAT" | awk '{print $1, $6}' | tr '\n' ' '; echo
