# Round-4 session 2: first run of the tiered builder -- PR tests, the one-shot tool, then the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s02
mkdir -p $O; rm -rf $O/*
timeout 600 python3 -m pytest tests -m gpu -x -q -k "pr" > $O/pytest_pr.txt 2>&1; tail -15 $O/pytest_pr.txt
GDN_PB_TRACE=1 timeout 300 python3 tools/pr_oneshot.py > $O/pr_oneshot.txt 2>&1; grep -v '^\[pb_order' $O/pr_oneshot.txt | tail -40
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.txt 2>&1; tail -15 $O/pytest_all.txt
