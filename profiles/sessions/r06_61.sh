# Round-6 session 61: counter traffic of the triangle count and of the BFS searches on the last code of the round
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 bash tools/traffic.sh r06t4 tc 23 2>&1 | tail -1
timeout 2400 bash tools/traffic.sh r06t5 bfs 27 2>&1 | tail -1
timeout 2400 bash tools/traffic.sh r06t6 bfs:1 27 2>&1 | tail -1
