# Round-3 session 25: which array's placement moves a PageRank iteration (tools/pr_place_probe.py), two processes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s25
mkdir -p $O; rm -rf $O/probe.txt
for p in 1 2; do
  echo "=== process $p" >> $O/probe.txt
  timeout 600 python3 tools/pr_place_probe.py 27 3 >> $O/probe.txt 2>&1
done
cat $O/probe.txt
