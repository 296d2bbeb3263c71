# Round-3 session 42: does the placement search pay below 2^28 edges (RMAT-24 / 25 plans; an eighth of RMAT-27 is 2^27.99)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s42
mkdir -p $O; rm -rf $O/*
for i in 1 2 3; do
for sc in 26; do
  for cfg in "GDN_PR_PLACE=0" "GDN_PLACE_MIN_EDGES=1 GDN_PR_PLACE_TRACE=1"; do
    echo "RMAT-$sc $cfg: $(env $cfg timeout 300 python3 tools/pr_notorch.py $sc 2 2>&1 | grep 'no-torch\|->' | sed 's/no-torch process: scale//' | tr '\n' ' ')" >> $O/t.txt
  done
done
done
cat $O/t.txt
