# Round-3 session 22: the whole GPU suite on the new bottom-up step, then the bench (BFS leg matters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s22
mkdir -p $O; rm -f $O/*
timeout 2400 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03s22/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
print(json.dumps(d.get("kernels", d.get("other_kernels", {})).get("bfs", {}))[:1500])
PY
