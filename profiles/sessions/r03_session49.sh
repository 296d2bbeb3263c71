# Round-3 session 49: PageRank bins of 2^13 rows (two phase-B workgroups per CU) on the four-tier layout, fresh processes interleaved
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s49
mkdir -p $O; rm -rf $O/*
for i in 1 2 3; do
  for cfg in "X=0" "GDN_PB_LOG_BIN=13"; do
    echo "$cfg: $(env $cfg timeout 600 python3 tools/pr_notorch.py 27 2 2>&1 | grep 'no-torch\|mid tiers\|hub tier' | tr '\n' ' ')" >> $O/t.txt
  done
done
cat $O/t.txt
