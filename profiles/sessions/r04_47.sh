# Round-4 session 47: PageRank mid-tier count again (interleaved records): 4 against 3 and 2, eight builds each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s47
mkdir -p $O; rm -rf $O/*
timeout 900 python3 tools/pr_ab_plan.py GDN_PB_MID 4 3 27 8 > $O/pr_mid3.txt 2>&1; tail -3 $O/pr_mid3.txt
timeout 900 python3 tools/pr_ab_plan.py GDN_PB_MID 4 2 27 4 > $O/pr_mid2.txt 2>&1; tail -3 $O/pr_mid2.txt
