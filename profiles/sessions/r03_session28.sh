# Round-3 session 28: placement tests, then the whole GPU suite and the bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s28
mkdir -p $O; rm -rf $O/*
timeout 600 python3 -m pytest tests -m gpu -q -x -k "placement" > $O/pytest_place.txt 2>&1; tail -15 $O/pytest_place.txt
timeout 2400 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
grep "passed\|failed" $O/pytest.txt | tail -3
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03s28/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["bfs"]["ms"], d["spmv"]["ms"], d["spmv"]["roofline"]["frac"])
PY
