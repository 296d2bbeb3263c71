# Round-4 session 30: SpMV plan on the tiered builder (values carried through the splits); BC with interleaved V
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s30
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_shapes.py -m gpu -q -x -k "spmv or pr_delta or bc" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt; grep -E "^FAILED|Error|assert" $O/pytest.txt | head -10
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > $O/pytest_fuzz.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz.txt; tail -5 $O/pytest_fuzz.txt | cut -c1-300
timeout 600 python3 -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "spmv" > $O/pytest_cfg.txt 2>&1; grep -E "passed|failed" $O/pytest_cfg.txt
GDN_PB_TRACE=1 timeout 600 python3 tools/spmv_ab_plan.py GDN_PB_BUILDER old new 25 2 > $O/ab.txt 2>&1; grep -v "^\[pb" $O/ab.txt | tail -8; grep "edges" $O/ab.txt | tail -2 | cut -c1-400
timeout 300 python3 tools/bc_plan.py 24 > $O/bc.txt 2>&1; tail -3 $O/bc.txt
GDN_PB_V_IL=0 timeout 300 python3 tools/bc_plan.py 24 > $O/bc_vil0.txt 2>&1; tail -3 $O/bc_vil0.txt
