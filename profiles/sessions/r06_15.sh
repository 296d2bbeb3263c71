# Round-6 session 15: the per-shard compute table on the final parts rule (four parts wherever a part holds half a round),
# the one-rank line, and the in-process multi-GPU tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s15
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 tools/shard_compute.py --n 1,2,4,8 --out $O/shard_compute.json > $O/shard.out 2> $O/shard.log; tail -1 $O/shard.out
Q="--no-extras --no-bfs --no-refsum --steps 20 --warmup 5"
timeout 600 python3 bench.py $Q --no-cpu > $O/plain.json 2> $O/plain.log
timeout 600 python3 bench.py --force-dist --gen range $Q --cpu-seconds 5 > $O/dist1.json 2> $O/dist1.log
timeout 900 python3 -m pytest tests/test_gpu_multi.py -x -q > $O/pytest_multi.txt 2>&1; tail -2 $O/pytest_multi.txt
python3 - <<'PY'
import json
O = "gpurun_out/r06s15"
for n in ("plain", "dist1"):
    r = json.loads([l for l in open("%s/%s.json" % (O, n)) if l.startswith("{")][-1])
    print(n, "ms/step %.3f" % r["ms_per_step"], [round(x, 3) for x in r["roofline"]["kernel_ms_parts"]], r.get("rccl_ranks"), r.get("collective_backend"), r["config"].get("parts"), (r.get("cpu_baseline") or {}).get("value"), (r.get("cpu_baseline") or {}).get("cores"))
PY
