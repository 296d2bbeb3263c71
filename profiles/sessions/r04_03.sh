# Round-4 session 3: where the tiered build spends its time (RMAT-22 under rocprofv3) and the RMAT-27 plan
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s03
mkdir -p $O; rm -rf $O/*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/pr_oneshot.py 22 > $O/pr_oneshot_rocprof.txt 2>&1
python3 - <<'PY' > $O/trace_top.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/r04s03/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:40]:
        print("%-60s calls %5s total %9.3f ms avg %9.4f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
head -40 $O/trace_top.txt
GDN_PB_TRACE=1 GDN_PR_PLACE=0 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-bfs > $O/bench_noplace.json 2> $O/bench_noplace.log
cat $O/bench_noplace.json; grep 'pb_build\|place' $O/bench_noplace.log | head
GDN_PB_TRACE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-bfs > $O/bench_place.json 2> $O/bench_place.log
cat $O/bench_place.json; grep 'pb_build\|place\]' $O/bench_place.log | tail -3
