# Round-5 session 15: the bottom-up wave kernel with 512-thread workgroups and <= 80 registers (24 waves per CU instead of 16): variant builds
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s15
mkdir -p $O; rm -rf $O/*
for v in "" bw512 bw512g; do
  if [ -n "$v" ]; then export GARDENIA_HIP_LIB=$GRAFT_REPO_ROOT/gardenia_amd/lib/var_$v/libgardenia_hip.so; else unset GARDENIA_HIP_LIB; fi
  timeout 300 python3 tools/bfs_ab.py 27 "" "" 2> $O/trace27_$v.txt | sed "s/^/lib=[$v] /" | tee -a $O/ab27.txt
  timeout 200 python3 tools/bfs_ab.py 24 "" "" 2> $O/trace24_$v.txt | sed "s/^/lib=[$v] /" | tee -a $O/ab24.txt
done
grep -A 10 "^== \[\] source 4" $O/trace27_bw512.txt | head -12
