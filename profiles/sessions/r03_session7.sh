# Round-3 session 7: non-R-MAT shapes, PageRank layout A/B (blocked vs merge-path)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s7
mkdir -p $O
timeout 900 python3 tools/shapes.py large $O/shapes_pb.json > $O/shapes_pb.log 2>&1
GDN_PR_LAYOUT=csr timeout 900 python3 tools/shapes.py large $O/shapes_csr.json > $O/shapes_csr.log 2>&1
python3 - <<'PY'
import json
for n in ("pb","csr"):
    j=json.load(open("gpurun_out/r03s7/shapes_%s.json"%n))
    for k,v in j.items():
        print(n,k,v["pagerank"])
PY
