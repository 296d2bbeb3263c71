# Round-5 session 48: counter traffic of the triangle count on the final code (K = 12288 at RMAT-23, packed walk bounds): tools/traffic.sh tc 23
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 bash tools/traffic.sh r05t3 tc 23 2>&1 | tail -1
