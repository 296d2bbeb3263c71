# Round-5 session 10: hashed frontier filter in front of the bottom-up scan's bitmap probes (A/B on one plan), BFS tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s10
mkdir -p $O; rm -rf $O/*
timeout 600 python3 tools/bfs_ab.py 27 "" "GDN_BFS_BU_FILTER=0" "GDN_BFS_BU_FILTER=4194304" "" > $O/ab27.txt 2> $O/ab27_trace.txt; cat $O/ab27.txt
timeout 300 python3 tools/bfs_ab.py 24 "" "GDN_BFS_BU_FILTER=0" "" > $O/ab24.txt 2> $O/ab24_trace.txt; cat $O/ab24.txt
timeout 300 python3 tools/bfs_ab.py 25 "" "GDN_BFS_BU_FILTER=0" > $O/ab25.txt 2> $O/ab25_trace.txt; cat $O/ab25.txt
grep -A 12 "^== \[\] source 5" $O/ab27_trace.txt | head -14
timeout 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_fuzz.py -x -q -m gpu -k "bfs or heads or random_graphs" > $O/t_bfs.txt 2>&1; tail -2 $O/t_bfs.txt
