# Round-3 session 26: the placement search of the PageRank / SpMV plans: fresh processes with and without, interleaved
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s26
mkdir -p $O; rm -rf $O/*
for i in 1 2 3 4; do
  for cfg in "GDN_PR_PLACE=0" "GDN_PR_PLACE_TRACE=1"; do
    echo "=== $cfg (process $i)" >> $O/place.txt
    env $cfg timeout 600 python3 tools/pr_notorch.py 27 2 2>&1 | grep "pr place\|no-torch\|crc" >> $O/place.txt
  done
done
grep -v "try" $O/place.txt
for i in 1 2 3; do
  for cfg in "GDN_SPMV_PLACE=0" "GDN_SPMV_PLACE_TRACE=1"; do
    echo "=== $cfg (process $i)" >> $O/place_spmv.txt
    env $cfg timeout 600 python3 tools/spmv_notorch.py 25 2>&1 | tail -12 >> $O/place_spmv.txt
  done
done
grep -v "try" $O/place_spmv.txt
