# Round-4 session 25: whole GPU suite + the default bench line with the lane-interleaved streams (PR, SpMV, SSSP)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s25
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; grep -E 'FAILED|passed|failed' $O/pytest_all.txt | head
python3 bench.py > $O/bench.json 2> $O/bench.log; python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04s25/bench.json"))
print("PR", d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_live_vertices"), d["roofline"]["kernel_ms_parts"], "plan", d["config"]["plan_build_s"])
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "physical_cores", "omp_proc_bind")} if d.get("cpu_baseline") else None)
print("parity", d.get("parity_note"))
print("bfs", d["bfs"]["ms_stats"], d["bfs"].get("ms_by_source"))
print("oneshot", d.get("pr_oneshot"))
print("spmv", d["spmv"]["kernel_ms"], d["spmv"]["roofline"]["frac"], d["spmv"]["plan_build_s"])
print("tc", d["tc"]["ms"], d["tc"]["plan_build_s"])
for k, v in d["traversal"].items():
    if isinstance(v, dict): print(k, v.get("ms"), v.get("plan_build_s"), v.get("oneshot_gdn_sssp_dev"))
PY
tail -5 $O/bench.log
