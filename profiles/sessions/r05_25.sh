# Round-5 session 25: the per-rank generation at the metric's size -- RMAT-27 through bench.py's N > 1 code path with this box's one GPU:
# (a) one rank, RCCL backend (--force-dist --gen range), (b) the same with the whole-graph build, (c) two ranks sharing the device
# (gloo collectives), every rank generating its own destination range
mkdir -p gpurun_out
timeout 900 python3 bench.py --force-dist --gen range --steps 20 --warmup 3 > gpurun_out/r05s25_range1.json 2> gpurun_out/r05s25_range1.log; tail -3 gpurun_out/r05s25_range1.log | cut -c1-300
timeout 900 python3 bench.py --force-dist --gen whole --steps 20 --warmup 3 --no-bfs --no-cpu --no-extras > gpurun_out/r05s25_whole1.json 2> gpurun_out/r05s25_whole1.log; tail -3 gpurun_out/r05s25_whole1.log | cut -c1-300
timeout 1200 python3 bench.py --gpus 2 --share-device --steps 10 --warmup 2 > gpurun_out/r05s25_range2.json 2> gpurun_out/r05s25_range2.log; tail -3 gpurun_out/r05s25_range2.log | cut -c1-300
python3 - <<'PY'
import json
for n in ("range1", "whole1", "range2"):
    try:
        r = json.loads([l for l in open("gpurun_out/r05s25_%s.json" % n) if l.startswith("{")][-1])
        print(n, "n_gpus", r["n_gpus"], "ms/step %.3f" % r["ms_per_step"], "edges", r["config"]["edges"], "build %.1f s" % r["graph_build_s"],
              "plan %.2f s" % r["config"]["plan_build_s"], "l1 %.17g" % r["pr_last_l1_change"], r["config"]["partition"][:150])
    except Exception as e:
        print(n, "failed:", e)
PY
