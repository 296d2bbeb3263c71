cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03s18
GDN_BFS_TRACE=1 timeout 600 python3 tools/bfs_notorch.py 27 > gpurun_out/r03s18/bfs_trace.txt 2>&1
tail -60 gpurun_out/r03s18/bfs_trace.txt | cut -c1-220
