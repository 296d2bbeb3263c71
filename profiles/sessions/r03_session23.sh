# Round-3 session 23: kernel times inside an SSSP solve (RMAT-24 U[1,255] delta 16, and unit weights)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s23
mkdir -p $O; rm -rf $O/*
export REPS=6
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rand -o sssp -- python3 tools/sssp_trace.py 24 16 rand plan > $O/rand.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/unit -o sssp -- python3 tools/sssp_trace.py 24 1 unit plan > $O/unit.log 2>&1
for k in rand unit; do
  f=$(find $O/$k -name "*kernel_stats.csv" | head -1)
  echo "== $k $f"; tail -3 $O/$k.log
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:16]:
    print("%-60s calls %5s total %9.1f us avg %8.1f us  %5.1f%%"%(r["Name"][:60],r["Calls"],float(r["TotalDurationNs"])/1e3,float(r["AverageNs"])/1e3,float(r["Percentage"])))
PY
done
