# Round-3 session 24b: SSSP tile padding rule at RMAT-25 / 26 / 27 (default = the new rule) against 32
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s24
mkdir -p $O; rm -rf $O/sssp_b.txt
export REPS=5
for sc in 25 26 27; do
for cfg in "X=0" "GDN_SSSP_PAD=32" "GDN_SSSP_PAD=64"; do
  echo "=== RMAT-$sc $cfg" >> $O/sssp_b.txt
  env $cfg timeout 300 python3 tools/sssp_trace.py $sc 16 rand plan 2>&1 | grep "RMAT" | awk '{print $6}' | tr '\n' ' ' >> $O/sssp_b.txt
  echo >> $O/sssp_b.txt
done
done
cat $O/sssp_b.txt
