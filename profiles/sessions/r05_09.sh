# Round-5 session 9: bottom-up from an edge share of 1/8 on skewed graphs (new default) against 1/3: RMAT-27, 24, 22, uniform26, the stand-in shapes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s09
mkdir -p $O; rm -rf $O/*
timeout 600 python3 tools/bfs_ab.py 27 "" "GDN_BFS_BU_EDGE_DIV=3" "" > $O/ab27.txt 2> $O/ab27_trace.txt; cat $O/ab27.txt; grep "plan: heads" $O/ab27_trace.txt | head -2
timeout 300 python3 tools/bfs_ab.py 24 "" "GDN_BFS_BU_EDGE_DIV=3" "GDN_BFS_BU_EDGE_DIV=8" > $O/ab24.txt 2> $O/ab24_trace.txt; cat $O/ab24.txt; grep "plan: heads" $O/ab24_trace.txt | head -1
timeout 300 python3 tools/bfs_ab.py 25 "" "GDN_BFS_BU_EDGE_DIV=3" > $O/ab25.txt 2> $O/ab25_trace.txt; cat $O/ab25.txt; grep "plan: heads" $O/ab25_trace.txt | head -1
timeout 400 python3 tools/bfs_shapes_trace.py uniform26 > $O/bfs_uniform26.txt 2> $O/bfs_uniform26_trace.txt; head -3 $O/bfs_uniform26.txt; grep "plan: heads" $O/bfs_uniform26_trace.txt | head -1
timeout 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py -x -q -m gpu -k "bfs" > $O/t_bfs.txt 2>&1; tail -2 $O/t_bfs.txt
