# Round-5 session 37: on the code of the BFS changes (compact records, deferred depths): the default bench line + kernel statistics
# (tools/profile_r05.sh without its PMC passes: the PageRank kernels did not change) and the BFS counter traffic again (tools/traffic.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 bash tools/profile_r05.sh > gpurun_out/r05_profile.log 2>&1; tail -3 gpurun_out/r05_profile.log
timeout 2400 bash tools/traffic.sh r05t2 bfs 27 2>&1 | tail -1
tail -c 1500 gpurun_out/r05/bench.json | head -c 600; echo
