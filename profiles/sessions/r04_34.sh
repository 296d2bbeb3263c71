# Round-4 session 34: kernel times of the binned top-down level (old form) under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s34
mkdir -p $O; rm -rf $O/*
GDN_BFS_BTD_FORM=old rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bfs_notorch.py 27 > $O/bfs.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04s34/trace/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "bfs" in n or "btd" in n: print(n[:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
