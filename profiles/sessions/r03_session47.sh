# Round-3 session 47: kernel times of the sweeps with record tiers (floor 256) and without
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s47
mkdir -p $O; rm -rf $O/*
for cfg in "GDN_SSSP_TIERS=0" "GDN_SSSP_TIER_MIN_DEG=256" "GDN_SSSP_TIER_MIN_DEG=64"; do
  tag=$(echo $cfg | tr '=' '_')
  env $cfg REPS=4 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o s -- python3 tools/sssp_trace.py 24 16 rand plan > $O/$tag.log 2>&1
  echo "== $cfg"; grep RMAT $O/$tag.log | tail -1
  python3 - "$O/$tag/s_kernel_stats.csv" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "sssp_pb" in n or "tier_gather" in n: print("   %-58s calls %4s avg %8.1f us"%(n.split("(")[0][:58],r["Calls"],float(r["AverageNs"])/1e3))
PY
done
