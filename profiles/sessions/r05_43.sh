# Round-5 session 43: TC with the core's size by graph size: the TC tests, the configs tests that name K, the bench's tc + standins blocks
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tc_" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "triangle or config4 or skewed" 2>&1 | tail -3
timeout 900 python3 bench.py --steps 5 --warmup 2 --no-bfs --no-cpu > gpurun_out/r05s43_bench.json 2> gpurun_out/r05s43_bench.log
python3 - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r05s43_bench.json") if l.startswith("{")][-1])
t = r["tc"]; o = r["standins"]["tc_orkut_like"]
print("tc RMAT-23:", t["ms"], "plan", t["plan_build_s"], "oneshot", t["oneshot_gdn_tc_dev"][:1])
print("tc orkut-like:", {k: o[k] for k in o if k in ("ms", "plan_build_s", "core_ranks", "triangles")})
PY
