# Round-6 session 40: counter traffic of the BFS search on the final code (fast source and the slow one)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 bash tools/traffic.sh r06t2 bfs 27 2>&1 | tail -1
timeout 2400 bash tools/traffic.sh r06t3 bfs:1 27 2>&1 | tail -1
