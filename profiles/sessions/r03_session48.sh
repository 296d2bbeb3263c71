# Round-3 session 48: SSSP record tiers at the default floor (nbins / 4): scales 24 / 25 / 26, both weight kinds, with / without; tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s48
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests -m gpu -q -x -k "sssp or SSSP" > $O/pytest.txt 2>&1; grep "passed\|failed" $O/pytest.txt | tail -2
for sc in 24 25 26; do
for cfg in "GDN_SSSP_TIERS=0" "X=0"; do
  for w in "16 rand" "1 unit"; do
    echo "RMAT-$sc $cfg | $w: $(env $cfg REPS=4 timeout 300 python3 tools/sssp_trace.py $sc $w plan 2>&1 | grep 'RMAT' | awk '{print $6}' | tr '\n' ' ')" >> $O/t.txt
  done
done
done
cat $O/t.txt
