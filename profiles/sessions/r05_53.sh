# Round-5 session 53: the BFS plan's size thresholds checked at RMAT-26 (outer hubs from 2^26 vertices, compact records and deferred depths from 2^25)
mkdir -p gpurun_out
timeout 900 python3 tools/bfs_ab.py 26 "" "GDN_BFS_HUBS2=0" "GDN_BFS_REC_COMPACT=0" "GDN_BFS_DEFER_DEPTH=0" "GDN_BFS_TD_DEFER_MIN=100000000000" "" 2> gpurun_out/r05s53_trace_26.txt | tee gpurun_out/r05s53_ab.txt
