# Round-5 session 30: deferred depths with the finish pass on static indices: A/B at RMAT-27 / 25 / 24, uniform 2^26, then parity at full size
mkdir -p gpurun_out
for s in 27 25 24; do
  timeout 600 python3 tools/bfs_ab.py $s "GDN_BFS_DEFER_DEPTH=0" "GDN_BFS_DEFER_DEPTH=1" "GDN_BFS_DEFER_DEPTH=0" "GDN_BFS_DEFER_DEPTH=1" 2> gpurun_out/r05s30_trace_$s.txt | tee -a gpurun_out/r05s30_ab.txt
done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "bfs" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs" 2>&1 | tail -4
