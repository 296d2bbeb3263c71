# Round-5 session 46: the pass that writes the kept levels' depths started early, beside the light levels at the end of a search (second stream):
# parity + fuzz + full size, then the A/B (GDN_BFS_FINISH_EARLY=0: the whole pass at the end)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs or sssp_equal" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "heads or deferred or plans" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "bfs or sssp" 2>&1 | tail -3
for s in 27 25 u26; do
  timeout 600 python3 tools/bfs_ab.py $s "GDN_BFS_FINISH_EARLY=0" "" "GDN_BFS_FINISH_EARLY=0" "" 2> gpurun_out/r05s46_trace_$s.txt | tee -a gpurun_out/r05s46_ab.txt
done
