# Round-3 session 5: PageRank plan knobs re-measured on the 4-tier layout (interleaved repetitions, fresh process each),
# counters of the shipped TC kernel, SSSP default after the binned passes went opt-in.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s5
mkdir -p $O
( date -u +"%Y-%m-%dT%H:%M:%SZ"; rocminfo 2>/dev/null | grep -m1 -i "uuid.*GPU" ) > $O/session.txt 2>&1
for rep in 1 2 3; do
for cfg in "BASE=1" "GDN_PB_PAD=16 GDN_PB_LOG_GROUP=4" "GDN_PB_HUB_ROWS=1" "GDN_PB_MID_MIN16=2" "GDN_PB_V8=1" "GDN_PB_MID=5"; do
  echo "=== $cfg rep $rep" >> $O/pr_knobs.txt
  env $cfg timeout 300 python3 tools/pr_notorch.py 27 2 2>&1 | grep "no-torch\|mid tiers" >> $O/pr_knobs.txt
done
done
grep "===\|no-torch" $O/pr_knobs.txt | paste - - | awk '{print $2,$3,$4,$5, $10,$11,$13,$14,$16,$17}'
for kind in rand unit; do
  d=16; [ $kind = unit ] && d=1
  REPS=6 timeout 300 python3 tools/sssp_trace.py 24 $d $kind plan >> $O/sssp_default.txt 2>&1
done
cat $O/sssp_default.txt
bash tools/pmc_generic.sh tc3 tc_count tools/tc_notorch.py 21 1 > $O/tc_pmc.txt 2>&1
cat $O/tc_pmc.txt
rm -rf gpurun_out/pmc_tc3
