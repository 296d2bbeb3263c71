# Round-4 session 46: do cheaper (interleaved) record streams move the optimum towards more tier edges?  SSSP tier floor, PageRank tier count
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s46
mkdir -p $O; rm -rf $O/*
for D in 64 32; do timeout 600 python3 tools/sssp_ab_plan.py GDN_SSSP_TIER_MIN_DEG 128 $D 24 2 2>&1 | grep -v round | grep median; done > $O/sssp_floor.txt; cat $O/sssp_floor.txt
timeout 600 python3 tools/pr_ab_plan.py GDN_PB_MID 4 5 27 3 > $O/pr_mid5.txt 2>&1; tail -3 $O/pr_mid5.txt
timeout 600 python3 tools/pr_ab_plan.py GDN_PB_MID 4 3 27 3 > $O/pr_mid3.txt 2>&1; tail -3 $O/pr_mid3.txt
