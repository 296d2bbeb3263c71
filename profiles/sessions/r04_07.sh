# Round-4 session 7: poisoned scratch -- which paths count on zeroed memory
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s07
mkdir -p $O; rm -rf $O/*
for v in "old" "new" ; do
  for pool in 1 0; do
    echo "== order $v pool $pool poisoned"
    GDN_SCRATCH_POISON=1 GDN_SCRATCH_POOL=$pool python3 tools/debug/pr_seed.py 200007 $v 2>&1 | grep -v '^\[pb\|^  \[pb' | tail -4
  done
done > $O/seed.txt 2>&1
cat $O/seed.txt
GDN_SCRATCH_POISON=1 timeout 1200 python3 -m pytest tests -m gpu -q > $O/pytest_poison.txt 2>&1; grep -E 'FAILED|passed|failed' $O/pytest_poison.txt | head -40
