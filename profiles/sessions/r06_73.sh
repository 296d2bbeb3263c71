# Round-6 session 73 (last): the whole GPU suite and the full default bench line on the last library of the round (core kernel with two rows per group)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s73
mkdir -p $O; rm -rf $O/*
timeout 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 2400 python3 -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_all.txt 2>&1; grep -E "FAILED|passed|failed|Error" $O/pytest_all.txt | head; grep -E "s call" $O/pytest_all.txt | head -8
timeout 1500 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log; tail -2 $O/bench.log | cut -c1-300
python3 - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06s73/bench.json") if l.startswith("{")][-1])
print("ms/step %.3f frac %.3f" % (r["ms_per_step"], r["roofline"]["frac"]), "refsum", {k: r["pr_reference_sum"][k] for k in ("ms_per_step", "frac", "rows_resummed", "longest_row")} if r.get("pr_reference_sum") else None)
print("bfs", r["bfs"]["ms_by_source"], r["bfs"].get("gteps_on_the_reference_timer"), r["bfs"].get("depth_finish_pass_ms"), r["bfs"].get("unreached_fill_modelled_ms"))
print("spmv", r["spmv"]["ms"], r["spmv"]["roofline"]["frac"]); print("tc", r["tc"]["ms"]); print("oneshot", r["pr_oneshot"]["pb"], r["pr_oneshot"]["auto"])
print("lj", r["standins"]["pr_lj_like"]["kernel_ms"], r["standins"]["pr_lj_like"]["roofline"]["frac"], "orkut", r["standins"]["tc_orkut_like"]["ms"])
print("trav", {k: v.get("ms") for k, v in r["traversal"].items() if isinstance(v, dict)})
print("cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], "parity", r["parity_note"]["rows_beyond_tolerance"], r["parity_note"].get("converged", {}).get("rows_beyond_tolerance_converged"))
PY
