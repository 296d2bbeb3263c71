# Round-5 session 26: BFS with the source's degree read in front of the fill; the share of the per-search initialisation (GDN_BFS_TIME_INIT);
# the bench line's new fields at small scale
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench_sharded.py -x -q -m gpu -k "carries" 2>&1 | tail -4
timeout 600 python3 tools/bfs_ab.py 27 "" "GDN_BFS_TIME_INIT=1" "" 2> gpurun_out/r05s26_trace.txt | tee gpurun_out/r05s26_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs" 2>&1 | tail -3
