# Round-4 session 48: kernel times of the two dense bottom-up passes (experiment, GDN_BFS_BU_FORM=d) under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s48
mkdir -p $O; rm -rf $O/*
GDN_BFS_BU_FORM=d rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/bfs_notorch.py 27 > $O/bfs.txt 2>&1
grep "BFS RMAT" $O/bfs.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04s48/trace/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "bfs_bu" in r["Kernel_Name"] or "bfs_bud" in r["Kernel_Name"]]
for r in rows[:14]:
    print(r["Kernel_Name"][:30], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0, "us")
PY
