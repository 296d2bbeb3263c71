# Round-5 session 16: BFS tests on the 512-thread bottom-up kernel (heads forced on small graphs, fuzz with heads), RMAT-27 oracle test, A/B again
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s16
mkdir -p $O; rm -rf $O/*
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu -k "bfs or heads or random_graphs or bc" > $O/t_bfs.txt 2>&1; tail -2 $O/t_bfs.txt
timeout 300 python3 tools/bfs_ab.py 27 "" "" > $O/ab27.txt 2> $O/trace27.txt; cat $O/ab27.txt
timeout 300 python3 tools/bfs_ab.py 25 "" > $O/ab25.txt 2> $O/trace25.txt; cat $O/ab25.txt
