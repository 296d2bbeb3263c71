# Round-5 session 3: new tests (placement search with fresh vals, SpMV option, stand-ins, 8-rank bench), the whole bench line with the search
# trace, BFS on hub-less shapes per engine
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s03
mkdir -p $O; rm -rf $O/*
timeout 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "placement or oneshot_on_the_blocked or reference_sum" > $O/t_parity.txt 2>&1; tail -3 $O/t_parity.txt
timeout 600 python3 -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "rmat_build_ex or standin" -s > $O/t_standin.txt 2>&1; grep -E "LJ-like|Orkut-like|passed|failed|Error|assert" $O/t_standin.txt | head
timeout 900 python3 -m pytest tests/test_gpu_bench_sharded.py -x -q -m gpu -k "eight or carries" > $O/t_bench.txt 2>&1; tail -5 $O/t_bench.txt
GDN_PR_PLACE_TRACE=1 timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; grep -E "place\]|converged parity|stand-ins|cpu baseline" $O/bench.err | cut -c1-400 | tail -40; python3 - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05s03/bench.json") if l.startswith("{")][-1])
    print("PR ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "parts", d["roofline"]["kernel_ms_parts"], "plan_build_s", d["config"]["plan_build_s"])
    print("bfs", d["bfs"]["ms_by_source"] if "bfs" in d else None)
    print("standins", json.dumps(d.get("standins"))[:1500])
    print("spmv oneshot", d["spmv"].get("oneshot_gdn_spmv"), d["spmv"].get("oneshot_gdn_spmv_blocked_layout"))
    print("tc", d["tc"]["ms"], d["tc"].get("oneshot_gdn_tc_dev"))
    print("parity", json.dumps(d.get("parity_note"))[:1200])
except Exception as e:
    print("bench parse failed", e)
PY
timeout 400 python3 tools/bfs_shapes_trace.py uniform small_world > $O/bfs_shapes.txt 2> $O/bfs_shapes_trace.txt; cat $O/bfs_shapes.txt
