# Round-4 session 45: one-shot SSSP layout build: po_tiles ranks tier records by LDS atomics; parity + phases of the build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s45
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "sssp" > $O/pytest_sssp.txt 2>&1; grep -E "passed|failed" $O/pytest_sssp.txt
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "800001" > $O/pytest_fuzz.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz.txt
GDN_PB_TRACE=1 timeout 300 python3 tools/sssp_prep.py 24 > $O/prep.txt 2>&1; grep -E "pb_build_out\]|scale 24" $O/prep.txt | tail -24
timeout 300 python3 tools/sssp_prep.py 24 > $O/prep_untraced.txt 2>&1; grep -E "scale 24" $O/prep_untraced.txt
