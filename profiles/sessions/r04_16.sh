# Round-4 session 16: kernel times of CC without the reverse graph
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s16
mkdir -p $O; rm -rf $O/*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/cc_notorch.py 24 > $O/cc.txt 2>&1
python3 - <<'PY' > $O/trace_top.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/r04s16/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:24]:
        print("%-60s calls %5s total %9.3f ms avg %9.4f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
cat $O/trace_top.txt
