# Round-6 session 34: where the bottom-up step of the slow BFS source is (level 3 of source 5): compile-time ablations of bfs_bu_wave_kernel's second stage
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s34
mkdir -p $O; rm -rf $O/*
for v in base babl1 babl2 babl3; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  timeout 600 python3 tools/bfs_runs.py 27 2 1 > $O/bfs_$v.txt 2>&1; echo "== $v"; grep -E "level [2345] (bottom|top)" $O/bfs_$v.txt | head -12
done
