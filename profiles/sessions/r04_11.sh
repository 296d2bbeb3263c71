# Round-4 session 11: where gdn_sssp_dev's preparation goes (RMAT-24), and the bench line's one-shot blocks
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s11
mkdir -p $O; rm -rf $O/*
GDN_PB_TRACE=1 python3 tools/sssp_prep.py 24 > $O/sssp_prep.txt 2>&1; grep -v '^\[' $O/sssp_prep.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/sssp_prep.py 24 > $O/sssp_prep_rocprof.txt 2>&1
python3 - <<'PY' > $O/trace_top.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/r04s11/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:32]:
        print("%-60s calls %5s total %9.3f ms avg %9.4f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
head -32 $O/trace_top.txt
