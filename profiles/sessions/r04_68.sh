# Round-4 session 68: after the candidate-width fix: the failing seed, the new test, SSSP parity, and the rest of the "heads" sweep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s68
mkdir -p $O; rm -rf $O/*
export OMP_NUM_THREADS=4
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_fullsize.py -m gpu -q -x -k "sssp" > $O/pytest_sssp.txt 2>&1; grep -E "passed|failed" $O/pytest_sssp.txt
( FUZZ_PLANS=1 GDN_BFS_HEADS_MIN_NNZ=1 GDN_BFS_HUB_MIN=0 GDN_BFS_BU_EDGE_DIV=1000000000 GDN_BFS_BTD=0 GDN_SSSP_TIER_MIN_NNZ=1 GDN_SSSP_TIER_MIN_DEG=2 GDN_SSSP_DENSE_IN=100000 timeout 1500 python3 tests/aids/fuzz_parity.py 1100 6000914 > $O/heads.txt 2>&1; grep -E "borderline|MISMATCH|fuzz parity" $O/heads.txt ) &
( FUZZ_PLANS=1 GDN_SSSP_DENSE_IN=100000 GDN_SSSP_ADAPT_AFTER=0 timeout 1500 python3 tests/aids/fuzz_parity.py 800 9000001 > $O/early_sweeps.txt 2>&1; grep -E "borderline|MISMATCH|fuzz parity" $O/early_sweeps.txt ) &
wait
