# Round-4 session 113: the whole GPU suite under the allocation fence on the final code
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s113
mkdir -p $O; rm -rf $O/*
GDN_ALLOC_FENCE=1 timeout 2400 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/suite_fence.txt 2>&1; grep -E 'FAILED|passed|failed|Memory access' $O/suite_fence.txt | head -5
