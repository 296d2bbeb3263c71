# Round-4 session 55 (rows whose head is their only in-neighbour are not queued for a scan): the bottom-up step with the wave as the unit (bfs_bu_wave_kernel): parity, BFS RMAT-27 / 24 A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s55
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shapes.py tests/test_gpu_configs.py -m gpu -q -x -k "bfs or bc" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt; grep -E "^FAILED|Error" $O/pytest.txt | head -5
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "800001 or 300001" > $O/pytest_fuzz.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz.txt; tail -3 $O/pytest_fuzz.txt | cut -c1-300
GDN_BFS_TRACE=1 timeout 600 python3 tools/bfs_notorch.py 27 > $O/bfs_new.txt 2>&1; grep -E "bottom-up|BFS RMAT" $O/bfs_new.txt | grep -v "hubs in" | head -16
timeout 600 python3 tools/bfs_notorch.py 27 > $O/bfs_new_untraced.txt 2>&1; grep -E "BFS RMAT" $O/bfs_new_untraced.txt
GDN_BFS_BU_FORM=window timeout 600 python3 tools/bfs_notorch.py 27 > $O/bfs_old_untraced.txt 2>&1; grep -E "BFS RMAT" $O/bfs_old_untraced.txt
timeout 600 python3 tools/bfs_notorch.py 24 > $O/bfs_new_24.txt 2>&1; grep -E "BFS RMAT" $O/bfs_new_24.txt
GDN_BFS_BU_FORM=window timeout 600 python3 tools/bfs_notorch.py 24 > $O/bfs_old_24.txt 2>&1; grep -E "BFS RMAT" $O/bfs_old_24.txt
