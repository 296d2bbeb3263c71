# Round-6 session 74: counter traffic of the triangle count on the last library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 bash tools/traffic.sh r06t7 tc 23 2>&1 | tail -1
