# Round-3 session 50: SSSP record tiers at small scales (threshold 2^22 edges): RMAT-19 .. 22 with / without
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s50
mkdir -p $O; rm -rf $O/*
for sc in 19 20 21 22; do
for cfg in "GDN_SSSP_TIERS=0" "X=0"; do
  echo "RMAT-$sc $cfg: $(env $cfg REPS=5 timeout 300 python3 tools/sssp_trace.py $sc 16 rand plan 2>&1 | grep 'RMAT' | awk '{print $6}' | tr '\n' ' ') | unit $(env $cfg REPS=5 timeout 300 python3 tools/sssp_trace.py $sc 1 unit plan 2>&1 | grep 'RMAT' | awk '{print $6}' | tr '\n' ' ')" >> $O/t.txt
done
done
cat $O/t.txt
