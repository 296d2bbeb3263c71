# Round-4 session 18: per-level trace of the three BFS sources on RMAT-27; new tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s18
mkdir -p $O; rm -rf $O/*
GDN_BFS_TRACE=1 python3 tools/bfs_notorch.py 27 > $O/bfs_trace.txt 2>&1; grep -v '^$' $O/bfs_trace.txt | tail -60
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_shot or oriented_input" > $O/pytest.txt 2>&1; grep -E 'FAILED|passed|failed|Error' $O/pytest.txt | head
