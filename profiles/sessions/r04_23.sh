# Round-4 session 23: V in lane-interleaved blocks of 512 edges (phase B main stream): parity, A/B on RMAT-27 / 24
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s23
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q -x -k "pr or pagerank or PageRank or PR" > $O/pytest_pr.txt 2>&1; grep -E "passed|failed" $O/pytest_pr.txt
timeout 900 python3 -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "pr or pagerank or PageRank or PR" > $O/pytest_cfg.txt 2>&1; grep -E "passed|failed" $O/pytest_cfg.txt
timeout 600 python3 tools/pr_ab_plan.py GDN_PB_V_IL 0 1 27 5 > $O/ab_vil.txt 2>&1; tail -3 $O/ab_vil.txt
timeout 300 python3 tools/pr_ab_plan.py GDN_PB_V_IL 0 1 24 5 > $O/ab_vil_s24.txt 2>&1; tail -3 $O/ab_vil_s24.txt
