# Round-6 session 1: (a) the plain N = 1 line, (b) the N > 1 code path with one rank (--force-dist --gen range) under
# rocprofv3 --kernel-trace --stats and unprofiled: where the +0.85 ms of VERDICT r5 item 1 go, (c) the price of
# GDN_PR_SUM=reference on the rows of >= 10^4 in-edges, traced as well
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s01
mkdir -p $O; rm -rf $O/*
Q="--no-cpu --no-extras --no-bfs --steps 20 --warmup 5"
timeout 600 python3 bench.py $Q > $O/plain.json 2> $O/plain.log
timeout 600 python3 bench.py --force-dist --gen range $Q > $O/dist1.json 2> $O/dist1.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_dist -- python3 bench.py --force-dist --gen range $Q > $O/dist1_rocprof.json 2> $O/trace_dist.log
GDN_PR_SUM=reference GDN_PR_SUM_MIN_DEGREE=10000 timeout 600 python3 bench.py $Q > $O/refsum.json 2> $O/refsum.log
export GDN_PR_SUM=reference GDN_PR_SUM_MIN_DEGREE=10000
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ref -- python3 bench.py $Q > $O/refsum_rocprof.json 2> $O/trace_ref.log
unset GDN_PR_SUM GDN_PR_SUM_MIN_DEGREE
python3 - <<'PY'
import json, glob, csv
O = "gpurun_out/r06s01"
for n in ("plain", "dist1", "dist1_rocprof", "refsum", "refsum_rocprof"):
    try:
        r = json.loads([l for l in open("%s/%s.json" % (O, n)) if l.startswith("{")][-1])
        print(n, "ms/step %.3f" % r["ms_per_step"], "kernel_ms %.3f" % r["roofline"]["kernel_ms"], r["roofline"]["kernel_ms_parts"], "step", r["step_ms"], "plan %.2f s" % r["config"]["plan_build_s"])
    except Exception as e:
        print(n, "failed:", e)
for t in ("trace_dist", "trace_ref"):
    for f in glob.glob("%s/%s/*/*_kernel_stats.csv" % (O, t)):
        print("==", t)
        for r in list(csv.DictReader(open(f)))[:14]:
            print("  %-70s calls %5s total %9.3f ms avg %8.4f ms" % (r["Name"].split("(")[0][:70], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
