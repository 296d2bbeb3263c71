# Round-3 session 4: GPU suite on the new defaults (4 mid tiers, binned SSSP passes, TC binary-search form, one-shot SSSP),
# SSSP traces with / without the binned passes, TC A/B, PageRank default, the full bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s4
mkdir -p $O
( date -u +"%Y-%m-%dT%H:%M:%SZ"; rocminfo 2>/dev/null | grep -m1 -i "uuid.*GPU" ) > $O/session.txt 2>&1
timeout 1800 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for kind in rand unit; do
  d=16; [ $kind = unit ] && d=1
  for nb in 0 1; do
    echo "=== RMAT-24 $kind delta $d NO_BINS=$nb" >> $O/sssp.txt
    if [ $nb = 1 ]; then export GDN_SSSP_NO_BINS=1; else unset GDN_SSSP_NO_BINS; fi
    GDN_SSSP_TRACE=1 timeout 300 python3 tools/sssp_trace.py 24 $d $kind plan >> $O/sssp.txt 2>&1
    REPS=8 timeout 300 python3 tools/sssp_trace.py 24 $d $kind plan >> $O/sssp.txt 2>&1
  done
done
unset GDN_SSSP_NO_BINS
grep "===\|RMAT-24\|binned\|sweep" $O/sssp.txt | cut -c1-200
for sc in 21 23; do
  for f in auto bs; do
    echo "=== TC RMAT-$sc form $f" >> $O/tc.txt
    if [ $f = bs ]; then export GDN_TC_FORM=bs; else unset GDN_TC_FORM; fi
    timeout 600 python3 tools/tc_notorch.py $sc >> $O/tc.txt 2>&1
  done
done
unset GDN_TC_FORM
cat $O/tc.txt | tail -20
for i in 1 2 3; do timeout 300 python3 tools/pr_notorch.py 27 2 2>&1 | grep "no-torch\|mid tiers" >> $O/pr_default.txt; done
cat $O/pr_default.txt
timeout 1500 python3 bench.py > $O/bench.json 2> $O/bench.log
tail -c 3000 $O/bench.json
