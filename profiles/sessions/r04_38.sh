# Round-4 session 38: SSSP: the tail list's out-degree sum left to the first tail pass: parity, timing on RMAT-24
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s38
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_fullsize.py -m gpu -q -x -k "sssp" > $O/pytest_sssp.txt 2>&1; grep -E "passed|failed" $O/pytest_sssp.txt
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > $O/pytest_fuzz.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz.txt
timeout 600 python3 tools/sssp_ab_plan.py GDN_SSSP_ADAPT 1 1 24 4 > $O/ab.txt 2>&1; grep -v round $O/ab.txt | tail -6
GDN_SSSP_TRACE=1 REPS=2 python3 tools/sssp_trace.py 24 16 rand > $O/trace.txt 2>&1; grep "sssp\]" $O/trace.txt | tail -9; grep RMAT $O/trace.txt
