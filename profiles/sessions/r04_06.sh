# Round-4 session 6: the fuzz seed that failed under the tiered builder: order of the variants, scratch pool on / off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s06
mkdir -p $O; rm -rf $O/*
for v in "old" "old,old,new,new" "new,old,old" ; do
  for pool in 1 0; do
    echo "== order $v pool $pool"
    GDN_SCRATCH_POOL=$pool python3 tools/debug/pr_seed.py 200007 $v 2>&1 | grep -v '^\[pb\|^  \[pb' | tail -8
  done
done > $O/seed.txt 2>&1
cat $O/seed.txt
