# Round-5 session 22: outer hubs of BFS's bottom-up step (GDN_BFS_HUBS2: ranks up to 2^21 tested against a rank-indexed 256 KB
# frontier bitmap instead of the 16 MB vertex-indexed one): parity first, then the A/B at three scales (plan rebuilt per set)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bfs" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "heads" 2>&1 | tail -5
for s in 27 25 24; do
  timeout 600 python3 tools/bfs_ab.py $s "GDN_BFS_HUBS2=0" "GDN_BFS_HUBS2=1" "GDN_BFS_HUBS2=0" "GDN_BFS_HUBS2=1" 2> gpurun_out/r05s22_trace_$s.txt | tee -a gpurun_out/r05s22_ab.txt
done
