# Round-6 session 27: the staging strip of the worklist pushes (one reservation on the queue's tail per flush, 11.3 ns each on one line): 256 / 512 / 1024 entries per wave -- BFS, SSSP, CC
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s27
mkdir -p $O; rm -rf $O/*
for v in base wl512 wl1024; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  timeout 600 python3 tools/bfs_runs.py 27 4 1 > $O/bfs_$v.txt 2>&1; grep -E "^round|level 3|level 2 top|traced:" $O/bfs_$v.txt
  for w in "sssp_u255 24" "sssp_unit 24" "cc 24" "bfs 24"; do
    timeout 600 python3 tools/traffic_run.py $w 6 > $O/${w%% *}_$v.txt 2>&1; echo $v: $(tail -1 $O/${w%% *}_$v.txt)
  done
done
