# Round-5 session 33: + the heavy top-down levels defer their depths (bitmap = visited ^ snapshot): fuzz, parity, full size, A/B of the threshold
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "heads or deferred or plans" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs" 2>&1 | tail -3
for s in 27 25 24 u26; do
  timeout 600 python3 tools/bfs_ab.py $s "GDN_BFS_TD_DEFER_MIN=100000000000" "" "GDN_BFS_TD_DEFER_MIN=100000000000" "" 2> gpurun_out/r05s33_trace_$s.txt | tee -a gpurun_out/r05s33_ab.txt
done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "bfs or sssp" 2>&1 | tail -4
