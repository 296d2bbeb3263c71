# Round-3 session 27: does the virtual address of `vals` tell a fast placement from a slow one?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s27
mkdir -p $O; rm -rf $O/*
for i in 1 2 3; do
  echo "=== process $i" >> $O/va.txt
  env GDN_PR_PLACE=1 GDN_PR_PLACE_OFFSETS=16 GDN_PR_PLACE_TRACE=1 timeout 600 python3 tools/pr_notorch.py 27 2 2>&1 | grep "pr place\|no-torch" >> $O/va.txt
done
grep "vals\|===\|->\|no-torch" $O/va.txt
