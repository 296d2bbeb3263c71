# Round-5 session 8: bottom-up scan with four in-neighbours per round trip (A/B on one plan, RMAT-27 and RMAT-24)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s08
mkdir -p $O; rm -rf $O/*
timeout 600 python3 tools/bfs_ab.py 27 "" "GDN_BFS_BU_SCAN=1" "GDN_BFS_BU_EDGE_DIV=8" "GDN_BFS_BU_EDGE_DIV=8,GDN_BFS_BU_SCAN=1" "" > $O/ab27.txt 2> $O/ab27_trace.txt; cat $O/ab27.txt
timeout 300 python3 tools/bfs_ab.py 24 "" "GDN_BFS_BU_SCAN=1" "" > $O/ab24.txt 2> $O/ab24_trace.txt; cat $O/ab24.txt
grep -A 12 "\[GDN_BFS_BU_EDGE_DIV=8\] source 5" $O/ab27_trace.txt | head -16
grep -A 12 "^== \[\] source 4" $O/ab27_trace.txt | head -14
timeout 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs" > $O/t_bfs.txt 2>&1; tail -2 $O/t_bfs.txt
