# Round-5 session 34: the finish pass with four trips' words in flight: A/B of the top-down deferral again, parity
mkdir -p gpurun_out
for s in 27 25 24 u26; do
  timeout 600 python3 tools/bfs_ab.py $s "GDN_BFS_TD_DEFER_MIN=100000000000" "" "GDN_BFS_TD_DEFER_MIN=100000000000" "" "GDN_BFS_DEFER_DEPTH=0" 2> gpurun_out/r05s34_trace_$s.txt | tee -a gpurun_out/r05s34_ab.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "heads or deferred" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "bfs" 2>&1 | tail -4
grep "written at the end" gpurun_out/r05s34_trace_27.txt | head -6
