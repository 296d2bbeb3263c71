# Round-6 session 42: kernel statistics of the BFS searches on the final code (three sources x 6 searches, untraced)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s42
mkdir -p $O; rm -rf $O/*
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bfs_runs.py 27 6 1 > $O/bfs.txt 2>&1
grep "^round" $O/bfs.txt
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r06s42/trace/*/*_kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if r["Name"].startswith("bfs_") or "mailbox" in r["Name"] or "Buffer" in r["Name"]]
    for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
        print("  %-40s calls %5s total %9.3f ms avg %8.4f ms" % (r["Name"].split("(")[0][:40], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
