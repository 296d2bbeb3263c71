# Round-4 session 33: binned top-down level with four ids per lane (bfs_btd_bin_big4_kernel): parity, BFS RMAT-27 by source
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s33
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shapes.py -m gpu -q -x -k "bfs or bc" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "300001" > $O/pytest_fuzz.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz.txt
GDN_BFS_TRACE=1 timeout 600 python3 tools/bfs_notorch.py 27 > $O/bfs_new.txt 2>&1; grep -E "binned|BFS RMAT" $O/bfs_new.txt | head -12
GDN_BFS_BTD_FORM=old GDN_BFS_TRACE=1 timeout 600 python3 tools/bfs_notorch.py 27 > $O/bfs_old.txt 2>&1; grep -E "binned|BFS RMAT" $O/bfs_old.txt | head -12
