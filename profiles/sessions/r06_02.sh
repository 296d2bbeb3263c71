# Round-6 session 2: the ticketed pull (gdn_pr_pull_parts_dev): its contract test, the sharded tests, then the N > 1 code path with
# one rank against the plain line at RMAT-27, and the per-shard compute tool (N = 2, 4, 8)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s02
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "ticketed or row_range or sharded or shards" > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 1500 python3 -m pytest tests/test_gpu_bench_sharded.py tests/test_gpu_multi.py -x -q > $O/pytest_b.txt 2>&1; tail -3 $O/pytest_b.txt
Q="--no-extras --no-bfs --steps 20 --warmup 5"
timeout 600 python3 bench.py $Q --no-cpu > $O/plain.json 2> $O/plain.log
timeout 600 python3 bench.py --force-dist --gen range $Q --cpu-seconds 5 > $O/dist1.json 2> $O/dist1.log
timeout 600 python3 bench.py --force-dist --gen range $Q --no-cpu > $O/dist1b.json 2> $O/dist1b.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_dist -- python3 bench.py --force-dist --gen range $Q --no-cpu > $O/dist1_rocprof.json 2> $O/trace_dist.log
timeout 1500 python3 tools/shard_compute.py --n 1,2,4,8 --out $O/shard_compute.json > $O/shard.out 2> $O/shard.log; tail -2 $O/shard.log | cut -c1-400
python3 - <<'PY'
import json, glob, csv
O = "gpurun_out/r06s02"
for n in ("plain", "dist1", "dist1b", "dist1_rocprof"):
    try:
        r = json.loads([l for l in open("%s/%s.json" % (O, n)) if l.startswith("{")][-1])
        print(n, "ms/step %.3f" % r["ms_per_step"], "kernel_ms %.3f" % r["roofline"]["kernel_ms"], r["roofline"]["kernel_ms_parts"], "step", r["step_ms"], "plan %.2f s" % r["config"]["plan_build_s"], r.get("rccl_ranks"), (r.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "failed:", e)
for f in glob.glob("%s/trace_dist/*/*_kernel_stats.csv" % O):
    for r in list(csv.DictReader(open(f)))[:12]:
        print("  %-70s calls %5s total %9.3f ms avg %8.4f ms" % (r["Name"].split("(")[0][:70], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
try:
    for s in json.load(open(O + "/shard_compute.json"))["shards"]:
        print("N %d rank %d edges %d bins %d wg/cu %.2f parts %d whole A %.3f B %.3f | ticketed A %.3f B %.3f wall %.3f | frac %.3f | model %s" % (
            s["n"], s["rank"], s["edges"], s["bins"], s["workgroups_per_cu"], s["parts"], s["whole_launch"]["phase_a_ms"], s["whole_launch"]["phase_b_ms"],
            s["ticketed"]["phase_a_ms"], s["ticketed"]["phase_b_ms"], s["ticketed"]["wall_ms"], s["frac_of_peak"],
            {k: round(v["predicted_step_ms"], 3) for k, v in s["xgmi_model"].items()}))
except Exception as e:
    print("shard_compute failed:", e)
PY
