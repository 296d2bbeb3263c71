# Round-5 session 49: the whole GPU suite under the allocation fence (GDN_ALLOC_FENCE=1: every device buffer ends at the end of its own block of
# whole pages and the scratch cache is off -- a kernel that leaves a buffer faults at once) on the final code
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s49
mkdir -p $O; rm -rf $O/*
GDN_ALLOC_FENCE=1 timeout 3000 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/suite_fence.txt 2>&1; grep -E 'FAILED|passed|failed|Memory access|Error' $O/suite_fence.txt | head -5
