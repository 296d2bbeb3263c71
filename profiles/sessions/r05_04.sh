# Round-5 session 4: stand-in tests again (policy trace), PageRank and BFS on BIG graphs of other shapes (device generator), search with the new order
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s04
mkdir -p $O; rm -rf $O/*
GDN_TRACE_POLICY=1 timeout 600 python3 -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "rmat_build_ex or standin or config2" -s > $O/t_standin.txt 2>&1; grep -E "policy|LJ-like|Orkut-like|passed|failed|Error|assert" $O/t_standin.txt | cut -c1-330 | head
timeout 600 python3 tools/pr_shape_big.py 26 16 0.25 0.25 0.25 26 16 0.45 0.22 0.22 26 16 0.45 0.22 0.22 flags=3 27 16 0.57 0.19 0.19 > $O/pr_shape_big.txt 2>&1; cat $O/pr_shape_big.txt | cut -c1-400
timeout 600 python3 tools/bfs_shapes_trace.py uniform26 > $O/bfs_uniform26.txt 2> $O/bfs_uniform26_trace.txt; cat $O/bfs_uniform26.txt; grep -A 14 "default" $O/bfs_uniform26_trace.txt | head -20
GDN_PR_PLACE_TRACE=1 timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-bfs > $O/bench_pr.json 2> $O/bench_pr.err; grep -E "place\] [0-9]|fresh 3|fresh 11" $O/bench_pr.err | tail -4; python3 -c "
import json; d=json.loads([l for l in open('$O/bench_pr.json') if l.startswith('{')][-1]); print('PR', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_parts'], d['config']['plan_build_s'])"
