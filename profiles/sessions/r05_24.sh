# Round-5 session 24: per-rank R-MAT generation (gdn_rmat_build_range + gdn_pr_squish_range + gdn_graph_pad_columns; bench.py --gen range,
# the default for N > 1): the range rows against the whole graph, then bench.py's 2- and 8-rank paths on one device, bit-equal with N = 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "rmat_build" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_bench_sharded.py -x -q -m gpu 2>&1 | tail -8
