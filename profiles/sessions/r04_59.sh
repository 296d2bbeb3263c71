# Round-4 session 59: top-down levels and the binned level's apply kernel take out-degrees from the head records: parity, timing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s59
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shapes.py tests/test_gpu_configs.py -m gpu -q -x -k "bfs or bc" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "800001 or 300001" > $O/pytest_fuzz.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz.txt
for i in 1 2; do timeout 600 python3 tools/bfs_notorch.py 27 2>&1 | grep "BFS RMAT"; done
