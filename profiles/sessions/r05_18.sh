# Round-5 session 18: counters of the bottom-up wave kernel (RMAT-25: passes of ~1 minute) -- where a bottom-up level spends its time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s18
mkdir -p $O; rm -rf $O/*
timeout 1500 bash tools/pmc_generic.sh bfs25 bfs_bu_wave tools/attic/bfs_notorch.py 25 > $O/pmc_bfs25.txt 2>&1; tail -40 $O/pmc_bfs25.txt
GDN_BFS_TRACE=1 timeout 120 python3 tools/attic/bfs_notorch.py 25 2>&1 | grep -E "level|BFS RMAT" | head -40 > $O/trace25.txt; head -14 $O/trace25.txt
