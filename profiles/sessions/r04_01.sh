# Round-4 session 1: what allocations cost, and the round-3 layout build as the baseline of the new builder
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s01
mkdir -p $O; rm -rf $O/*
tools/_bin/malloc_probe > $O/malloc_probe.txt 2>&1
GDN_PB_TRACE=1 python3 tools/pr_oneshot.py > $O/pr_oneshot.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/pr_oneshot.py > $O/pr_oneshot_rocprof.txt 2>&1
python3 - <<'PY' > $O/trace_top.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/r04s01/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:45]:
        print("%-60s calls %5s total %9.3f ms avg %9.4f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
cat $O/malloc_probe.txt; cat $O/pr_oneshot.txt | grep -v '^\[pb' ; head -50 $O/trace_top.txt
