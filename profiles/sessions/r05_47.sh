# Round-5 session 47: the early pass as a small grid on a low-priority stream: A/B of its size
mkdir -p gpurun_out
timeout 900 python3 tools/bfs_ab.py 27 "GDN_BFS_FINISH_EARLY=0" "" "GDN_BFS_FINISH_EARLY_BLOCKS=256" "GDN_BFS_FINISH_EARLY_BLOCKS=1024" "GDN_BFS_FINISH_EARLY_BLOCKS=2048" "GDN_BFS_FINISH_EARLY=0" "" 2> gpurun_out/r05s47_trace_27.txt | tee gpurun_out/r05s47_ab.txt
timeout 600 python3 tools/bfs_ab.py 25 "GDN_BFS_FINISH_EARLY=0" "" "GDN_BFS_FINISH_EARLY_BLOCKS=256" "GDN_BFS_FINISH_EARLY=0" "" 2> gpurun_out/r05s47_trace_25.txt | tee -a gpurun_out/r05s47_ab.txt
