# Round-6 session 14: reference-order sums through the stage pass (LDS slices -> row order) and streaming scans: parity, price
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s14
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "reference or ticketed or reserved" > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 1500 python3 -m pytest tests/test_gpu_configs.py -x -q -k "summation or rmat27" > $O/pytest_b.txt 2>&1; tail -3 $O/pytest_b.txt
Q="--no-extras --no-bfs --no-cpu --steps 20 --warmup 5"
timeout 900 python3 bench.py $Q > $O/plain.json 2> $O/plain.log
timeout 900 python3 bench.py $Q --refsum-min-degree 50000 > $O/plain_d50k.json 2> $O/plain_d50k.log
timeout 900 python3 bench.py $Q --refsum-min-degree 2000 > $O/plain_d2k.json 2> $O/plain_d2k.log
export GDN_PR_SUM=reference GDN_PR_SUM_MIN_DEGREE=10000
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ref -- python3 bench.py $Q --no-refsum > $O/refsum_rocprof.json 2> $O/trace_ref.log
unset GDN_PR_SUM GDN_PR_SUM_MIN_DEGREE
python3 - <<'PY'
import json, glob, csv
O = "gpurun_out/r06s14"
for n in ("plain", "plain_d50k", "plain_d2k", "refsum_rocprof"):
    try:
        r = json.loads([l for l in open("%s/%s.json" % (O, n)) if l.startswith("{")][-1])
        rs = r.get("pr_reference_sum") or {}
        print(n, "ms/step %.3f" % r["ms_per_step"], "plan %.2f" % r["config"]["plan_build_s"], "| refsum ms %.3f rows %s entries %s launches %s plan %.2f l1 %s vs %s" % (rs.get("ms_per_step", 0), rs.get("rows_resummed"), rs.get("entries_resummed"), rs.get("launches_per_iteration_for_the_resum"), rs.get("plan_build_s", 0), rs.get("pr_last_l1_change"), r["pr_last_l1_change"]))
    except Exception as e:
        print(n, "failed:", e)
for f in glob.glob("%s/trace_ref/*/*_kernel_stats.csv" % O):
    for r in list(csv.DictReader(open(f))):
        if "pr_ref" in r["Name"] or "pb_" in r["Name"]:
            print("  %-60s calls %5s total %9.3f ms avg %8.4f ms" % (r["Name"].split("(")[0][:60], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
