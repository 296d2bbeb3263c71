# Round-4 session 8: trace of the build under poison
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s08
mkdir -p $O; rm -rf $O/*
for part in head:65536 mid:24576:49152 tail:1; do
  echo "== $part"; GDN_PB_TRACE=1 GDN_SCRATCH_POISON_PART=$part GDN_SCRATCH_POISON=1 GDN_SCRATCH_POISON_SITES=1 python3 tools/debug/pr_seed.py 200007 new 2>&1 | grep '^new\|scratch\]\|rows with\|main edges\|edges 2783' | cut -c1-200
done > $O/seed.txt 2>&1
cat $O/seed.txt
