# Round-3 session 40: when the SSSP solve enters its sweeps (RMAT-24 / 26, U[1,255] delta 16 and unit weights)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s40
mkdir -p $O; rm -rf $O/*
export REPS=5
for cfg in "X=0" "GDN_SSSP_DENSE_IN=48" "GDN_SSSP_DENSE_IN=96" "GDN_SSSP_DENSE_IN=48 GDN_SSSP_DENSE_PRE=1000000" "GDN_SSSP_DENSE_IN=12" "X=0"; do
  for w in "24 16 rand" "24 1 unit" "26 16 rand"; do
    echo "$cfg | $w: $(env $cfg timeout 300 python3 tools/sssp_trace.py $w plan 2>&1 | grep 'RMAT' | awk '{print $6, $8}' | tr '\n' ' ')" >> $O/t.txt
  done
done
cat $O/t.txt
