# Round-5 session 59: the default bench line three times on one box with the stop ratio of the `vals` search at 0.90 (fresh processes), the placement tests
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "place" 2>&1 | tail -2
for i in 1 2 3; do
  timeout 900 python3 bench.py --no-cpu --no-extras --no-bfs > gpurun_out/r05s59_bench$i.json 2> gpurun_out/r05s59_bench$i.log
  python3 - gpurun_out/r05s59_bench$i.json <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("PR %.3f ms frac %.3f plan %.2f s parts %s" % (r["ms_per_step"], r["roofline"]["frac"], r["config"]["plan_build_s"], [round(x, 3) for x in r["roofline"]["kernel_ms_parts"]]))
PY
done
