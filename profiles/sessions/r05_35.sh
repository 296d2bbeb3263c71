# Round-5 session 35: grid of the finish pass (GDN_BFS_FINISH_BLOCKS) at RMAT-27
mkdir -p gpurun_out
timeout 900 python3 tools/bfs_ab.py 27 "" "GDN_BFS_FINISH_BLOCKS=1024" "GDN_BFS_FINISH_BLOCKS=2048" "GDN_BFS_FINISH_BLOCKS=4096" "GDN_BFS_FINISH_BLOCKS=16384" "GDN_BFS_FINISH_BLOCKS=65536" "" 2> gpurun_out/r05s35_trace_27.txt | tee gpurun_out/r05s35_ab.txt
grep "written at the end\|== \[" gpurun_out/r05s35_trace_27.txt | grep -A1 "source 4" | grep written
