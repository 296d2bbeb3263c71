# Round-4 session 35: binned top-down level: apply kernel with 16-byte id loads
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s35
mkdir -p $O; rm -rf $O/*
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shapes.py -m gpu -q -x -k "bfs or bc" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "300001" > $O/pytest_fuzz.txt 2>&1; grep -E "passed|failed" $O/pytest_fuzz.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bfs_notorch.py 27 > $O/bfs.txt 2>&1
grep "BFS RMAT" $O/bfs.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04s35/trace/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "btd" in n: print(n[:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
