# Round-5 session 40: the frontier listed behind the bitmap levels needs no read-back of its length (it is what the last level discovered): A/B, parity
mkdir -p gpurun_out
for s in 27 25 24 22; do
  timeout 600 python3 tools/bfs_ab.py $s "GDN_BFS_B2Q_READ=1" "" "GDN_BFS_B2Q_READ=1" "" 2> gpurun_out/r05s40_trace_$s.txt | tee -a gpurun_out/r05s40_ab.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "plans or heads or deferred" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "bfs" 2>&1 | tail -3
