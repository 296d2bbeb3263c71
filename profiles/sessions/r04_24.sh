# Round-4 session 24: SpMV with lane-interleaved V and mid-tier streams (records + factors): parity, A/B on RMAT-25
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s24
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q -x -k "spmv or pr_delta or pr or PR" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "200001" > $O/pytest_fuzz.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz.txt
timeout 600 python3 -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "spmv" > $O/pytest_cfg.txt 2>&1; grep -E "passed|failed" $O/pytest_cfg.txt
timeout 600 python3 tools/spmv_ab_plan.py GDN_PB_REC_IL,GDN_PB_V_IL 0 1 25 3 > $O/ab.txt 2>&1; tail -3 $O/ab.txt
timeout 600 python3 tools/spmv_ab_plan.py GDN_PB_V_IL 0 1 25 3 > $O/ab_v.txt 2>&1; tail -3 $O/ab_v.txt
