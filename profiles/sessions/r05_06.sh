# Round-5 session 6: SSSP BFS route + TC walked-elements tests, bench tests, the whole bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s06
mkdir -p $O; rm -rf $O/*
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "equal_weights or walked or placement" > $O/t_parity.txt 2>&1; tail -3 $O/t_parity.txt
timeout 600 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "sssp_unit" > $O/t_full.txt 2>&1; tail -2 $O/t_full.txt
timeout 600 python3 -m pytest tests/test_gpu_bench_sharded.py -x -q -m gpu -k "carries" > $O/t_bench.txt 2>&1; tail -3 $O/t_bench.txt
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05s06/bench.json") if l.startswith("{")][-1])
print("PR ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "parts", d["roofline"]["kernel_ms_parts"], "plan_build_s", d["config"]["plan_build_s"])
print("bfs", d["bfs"]["ms_by_source"], d["bfs"]["roofline"]["frac"], d["bfs"]["roofline"]["speed_vs_model"])
t = d["traversal"]
print("sssp_unit", t["sssp_unit"]["ms"], t["sssp_unit"].get("route"), t["sssp_unit"].get("ab_dense_sweeps"), t["sssp_unit"]["plan_build_s"], t["sssp_unit"]["roofline"])
print("sssp_u255", t["sssp_u1_255_delta16"]["ms"])
print("tc", d["tc"]["ms"], d["tc"]["roofline"]["frac"], d["tc"]["roofline"]["speed_vs_model"], d["tc"]["roofline"]["kernel_list_read_frac"], d["tc"]["roofline"].get("list_elements_walked"))
PY
