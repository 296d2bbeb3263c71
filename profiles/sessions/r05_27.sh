# Round-5 session 27: the bottom-up wave kernel on a COMPACT copy of the head records (rows with in-edges only): parity, then the A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "heads" 2>&1 | tail -3
for s in 27 25 24; do
  timeout 600 python3 tools/bfs_ab.py $s "GDN_BFS_REC_COMPACT=0" "GDN_BFS_REC_COMPACT=1" "GDN_BFS_REC_COMPACT=0" "GDN_BFS_REC_COMPACT=1" 2> gpurun_out/r05s27_trace_$s.txt | tee -a gpurun_out/r05s27_ab.txt
done
