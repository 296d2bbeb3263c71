# Round-5 session 32: deferred depths in every engine of the dense phase (binned top-down, dense sweep, bottom-up): fuzz + parity + full size, A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "heads or deferred or plans" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs" 2>&1 | tail -3
for s in 27 25 24 u26; do
  timeout 600 python3 tools/bfs_ab.py $s "GDN_BFS_DEFER_DEPTH=0" "GDN_BFS_DEFER_DEPTH=1" "GDN_BFS_DEFER_DEPTH=0" "GDN_BFS_DEFER_DEPTH=1" 2> gpurun_out/r05s32_trace_$s.txt | tee -a gpurun_out/r05s32_ab.txt
done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "bfs or sssp" 2>&1 | tail -4
