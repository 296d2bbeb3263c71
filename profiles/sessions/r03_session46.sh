# Round-3 session 46: SSSP record tiers: floor (out-edges per source) and number of tiers, RMAT-24 / 26
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s46
mkdir -p $O; rm -rf $O/*
for sc in 24 26; do
for cfg in "GDN_SSSP_TIERS=0" "X=0" "GDN_SSSP_TIER_MIN_DEG=32" "GDN_SSSP_TIER_MIN_DEG=64" "GDN_SSSP_TIER_MIN_DEG=256" "GDN_SSSP_TIER_MIN_DEG=512" "GDN_SSSP_TIERS=1" "GDN_SSSP_TIERS=2"; do
  echo "RMAT-$sc $cfg: $(env $cfg REPS=4 timeout 300 python3 tools/sssp_trace.py $sc 16 rand plan 2>&1 | grep 'RMAT' | awk '{print $6}' | tr '\n' ' ') | unit $(env $cfg REPS=4 timeout 300 python3 tools/sssp_trace.py $sc 1 unit plan 2>&1 | grep 'RMAT' | awk '{print $6}' | tr '\n' ' ')" >> $O/t.txt
done
done
cat $O/t.txt
for sc in 24 26; do env GDN_SSSP_TRACE=1 REPS=1 timeout 300 python3 tools/sssp_trace.py $sc 16 rand plan 2>&1 | grep "plan:" | head -1; done
