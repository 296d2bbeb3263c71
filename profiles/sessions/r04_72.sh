# Round-4 session 72: what the driver runs at round end: smoke(), pytest -m gpu -x, the default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s72
mkdir -p $O; rm -rf $O/*
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python3 -m pytest tests/ -x -q -m gpu > $O/pytest.txt 2>&1; tail -1 $O/pytest.txt
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')}); print(d['roofline']['frac'], d['cpu_baseline']['value'], d['gteps_bfs'])"
