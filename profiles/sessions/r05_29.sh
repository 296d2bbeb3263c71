# Round-5 session 29: deferred depths (the bottom-up levels leave the distance array alone; one sequential pass at the end writes the kept
# levels' depths and "unreached", no fill in front): parity, fuzz, full-size oracle parity, A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "heads" 2>&1 | tail -3
for s in 27 25; do
  timeout 600 python3 tools/bfs_ab.py $s "GDN_BFS_DEFER_DEPTH=0" "GDN_BFS_DEFER_DEPTH=1" "GDN_BFS_DEFER_DEPTH=0" "GDN_BFS_DEFER_DEPTH=1" 2> gpurun_out/r05s29_trace_$s.txt | tee -a gpurun_out/r05s29_ab.txt
done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "bfs" 2>&1 | tail -4
