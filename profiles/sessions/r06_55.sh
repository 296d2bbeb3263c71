# Round-6 session 55: the new TC defaults (hash-set kernel compiled for five waves per SIMD on dynamic LDS, queued first with the plan's items, four core workgroups per CU): tests, every graph, the one-shot call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s55
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "tc or triangle" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for g in 23 orkut 19 20 21 22 24; do timeout 600 python3 tools/tc_knob_ab.py $g 8 "" > $O/$g.txt 2>&1; tail -3 $O/$g.txt | head -2; done
timeout 900 python3 bench.py --no-cpu --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.log; python3 - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06s55/bench.json") if l.startswith("{")][-1])
print("tc", r["tc"]["ms"], r["tc"].get("oneshot_gdn_tc_dev"), "orkut", r["standins"]["tc_orkut_like"]["ms"], r["standins"]["tc_orkut_like"].get("oneshot_gdn_tc_dev"))
PY
