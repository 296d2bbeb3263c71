# Round-4 session 87: kernel times of the forward count with the core (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s87
mkdir -p $O; rm -rf $O/*
export GDN_TC_CORE_ONLY=8192
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o tc -- python3 tools/tc_core_ab.py 23 3 > $O/run.txt 2>&1
tail -3 $O/run.txt
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r04s87/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if "tc_" in r["Name"]:
            print(r["Name"][:70], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3, "total ms", float(r["TotalDurationNs"]) / 1e6)
PY
