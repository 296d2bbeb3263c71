# Round-4 session 22: lane-interleaved SSSP tier streams: parity, fuzz with tiers forced, A/B on RMAT-24
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s22
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sssp" > $O/pytest_sssp.txt 2>&1; grep -E "passed|failed" $O/pytest_sssp.txt
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "800001" > $O/pytest_fuzz.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz.txt
timeout 600 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "sssp" > $O/pytest_full.txt 2>&1; grep -E "passed|failed" $O/pytest_full.txt
timeout 600 python3 tools/sssp_ab_plan.py GDN_SSSP_REC_IL 0 1 24 3 > $O/ab.txt 2>&1; grep -v round $O/ab.txt | tail -8
