# Round-6 session 66: kernel statistics of betweenness centrality (resident plan, RMAT-24) on the last code
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s66
mkdir -p $O; rm -rf $O/*
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/attic/bc_notorch.py 24 plan > $O/bc.txt 2>&1; tail -5 $O/bc.txt
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r06s66/trace/*/*_kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if r["Name"].startswith("bc_") or r["Name"].startswith("void bc_") or "Buffer" in r["Name"] or "mailbox" in r["Name"] or "pb_" in r["Name"]]
    for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:16]:
        print("  %-44s calls %5s total %9.3f ms avg %8.4f ms" % (r["Name"].split("(")[0][:44], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
