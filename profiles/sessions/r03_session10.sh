# Round-3 session 10: SpMV record tiers (count and floor) re-measured with the linear-threshold picker
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s10
mkdir -p $O; rm -f $O/spmv_tiers.txt
for rep in 1 2 3; do
for cfg in "GDN_PB_MID=2 GDN_PB_MID_MIN16=4" "GDN_PB_MID=3 GDN_PB_MID_MIN16=4" "GDN_PB_MID=4 GDN_PB_MID_MIN16=4" "GDN_PB_MID=3 GDN_PB_MID_MIN16=1" "GDN_PB_MID=4 GDN_PB_MID_MIN16=1" "GDN_PB_MID=1 GDN_PB_MID_MIN16=4"; do
  echo "=== $cfg rep $rep" >> $O/spmv_tiers.txt
  env GDN_PB_TRACE=1 $cfg timeout 300 python3 tools/spmv_notorch.py 25 2>&1 | grep "spmv scale\|check\|pick_tiers" >> $O/spmv_tiers.txt
done
done
grep "===\|spmv scale" $O/spmv_tiers.txt | paste - - | sort
grep "check" $O/spmv_tiers.txt | sort | uniq -c
grep pick_tiers $O/spmv_tiers.txt | sort | uniq -c
