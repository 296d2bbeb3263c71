# Round-3 session 52: record + weight in one 8-byte word (one vector-memory instruction per record instead of two)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s52
mkdir -p $O; rm -rf $O/*
for rep in 1 2; do
for v in base rec64; do
  lib=gardenia_amd/lib/var_$v/libgardenia_hip.so
  [ $v = base ] && lib=gardenia_amd/lib/libgardenia_hip.so
  for sc in 24 26; do
    echo "$v RMAT-$sc: $(env GARDENIA_HIP_LIB=$PWD/$lib REPS=4 timeout 300 python3 tools/sssp_trace.py $sc 16 rand plan 2>&1 | grep 'RMAT' | awk '{print $6, $13}' | tr '\n' ' ')" >> $O/t.txt
  done
done
done
cat $O/t.txt
