# Round-4 session 39: SSSP host loop: one speculative split per bucket change; FAR length up to which the one-workgroup kernel changes buckets
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s39
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_fullsize.py -m gpu -q -x -k "sssp" > $O/pytest_sssp.txt 2>&1; grep -E "passed|failed" $O/pytest_sssp.txt
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > $O/pytest_fuzz.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz.txt
GDN_SSSP_SMALL=0 timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "100001" > $O/pytest_fuzz_host.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz_host.txt
for F in 65536 8192 2048 512; do echo "== GDN_SSSP_SMALL_FAR=$F"; GDN_SSSP_SMALL_FAR=$F timeout 600 python3 tools/sssp_ab_plan.py GDN_SSSP_ADAPT 1 1 24 2 2>&1 | grep -v round | grep -E "U\[|unit" | grep median | sort -u; GDN_SSSP_SMALL_FAR=$F timeout 300 python3 tools/sssp_delta_sweep.py uniform 23 8 2>&1 | head -1; done
