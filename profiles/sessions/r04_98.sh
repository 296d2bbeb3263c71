# Round-4 session 98: non-R-MAT shapes (large) with the TC core on by default; same with GDN_TC_CORE=0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s98
mkdir -p $O; rm -rf $O/*
timeout 1200 python3 tools/shapes.py large $O/shapes_large.json > $O/shapes.txt 2>&1; tail -3 $O/shapes.txt | cut -c1-300
GDN_TC_CORE=0 timeout 1200 python3 tools/shapes.py large $O/shapes_large_nocore.json > $O/shapes_nocore.txt 2>&1
python3 - <<'PY'
import json
a=json.load(open("gpurun_out/r04s98/shapes_large.json")); b=json.load(open("gpurun_out/r04s98/shapes_large_nocore.json"))
for k in a: print(k, "tc core", a[k].get("tc"), " no core", b[k].get("tc"))
PY
