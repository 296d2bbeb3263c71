# Round-4 session 40: kernel times of CC (RMAT-24, with / without the reverse graph) and of TC's plan build under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s40
mkdir -p $O; rm -rf $O/*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cc -- python3 tools/cc_notorch.py 24 > $O/cc.txt 2>&1
cat $O/cc.txt | tail -3
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04s40/cc/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "cc_" in n: print(n[:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
