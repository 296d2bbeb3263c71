# Round-5 session 55: fewer dispatches per BFS level -- the level counters zeroed by the kernel that reads them back, the hub count taken by the
# bottom-up workgroups themselves (no counter, no memset): parity, fuzz, full size, then timings at RMAT-27 / 25 / 24 / 22 (compare: sessions 40, 53)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs or sssp_equal" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "bfs or sssp" 2>&1 | tail -3
for s in 27 25 24 22; do
  timeout 600 python3 tools/bfs_ab.py $s "" "" 2> gpurun_out/r05s55_trace_$s.txt | tee -a gpurun_out/r05s55_ab.txt
done
