# Round-4 session 42: BC plans on the tiered builder: parity, plan build and solve times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s42
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_fullsize.py -m gpu -q -x -k "bc" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "300001 or 700001" > $O/pytest_fuzz.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz.txt
GDN_PB_TRACE=1 timeout 300 python3 tools/bc_plan.py 24 > $O/bc_new.txt 2>&1; grep -E "wall|BC plan" $O/bc_new.txt | cut -c1-200 | tail -4
GDN_PB_BUILDER=old GDN_PB_TRACE=1 timeout 300 python3 tools/bc_plan.py 24 > $O/bc_old.txt 2>&1; grep -E "wall|BC plan" $O/bc_old.txt | cut -c1-200 | tail -4
