# Round-3 session 24: SSSP sweep layout knobs at RMAT-24 / RMAT-26 (chunk = bin = 2^LOG ids, tile padding)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s24
mkdir -p $O; rm -rf $O/*
export REPS=5
for sc in 24 26; do
for cfg in "X=0" "GDN_SSSP_LOG=14" "GDN_SSSP_LOG=14 GDN_SSSP_PAD=64" "GDN_SSSP_LOG=13" "GDN_SSSP_LOG=15 GDN_SSSP_PAD=64" "X=0"; do
  echo "=== RMAT-$sc $cfg" >> $O/sssp.txt
  env $cfg timeout 300 python3 tools/sssp_trace.py $sc 16 rand plan 2>&1 | grep "RMAT" | awk '{print $6}' | tr '\n' ' ' >> $O/sssp.txt
  echo >> $O/sssp.txt
done
done
cat $O/sssp.txt
