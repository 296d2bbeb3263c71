# Round-6 session 47: kernel statistics of CC (with / without the reverse graph) and SSSP (unit, U[1,255]) at RMAT-24 on the final code
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s47
mkdir -p $O; rm -rf $O/*
for w in cc cc_out sssp_u255 sssp_unit; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- python3 tools/traffic_run.py $w 24 8 > $O/$w.txt 2>&1
  echo "== $w: $(tail -1 $O/$w.txt)"
  python3 - $w <<'PY'
import csv, glob, sys
w = sys.argv[1]
for f in glob.glob("gpurun_out/r06s47/%s/*/*_kernel_stats.csv" % w):
    rows = list(csv.DictReader(open(f)))
    for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:14]:
        if int(r["Calls"]) >= 8:
            print("  %-44s calls %5s per solve %6.1f x %8.4f ms = %7.3f ms" % (r["Name"].split("(")[0][:44], r["Calls"], int(r["Calls"]) / 8.0, float(r["AverageNs"]) / 1e6, int(r["TotalDurationNs"]) / 8e6))
PY
done
