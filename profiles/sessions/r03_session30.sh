# Round-3 session 30: placement spread of the SSSP plan (RMAT-24 / 26) and of the BFS plan
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s30
mkdir -p $O; rm -rf $O/*
timeout 300 python3 tools/sssp_replan.py 24 5 2>&1 | grep round > $O/sssp24.txt; cat $O/sssp24.txt
timeout 300 python3 tools/sssp_replan.py 26 3 2>&1 | grep round > $O/sssp26.txt; cat $O/sssp26.txt
