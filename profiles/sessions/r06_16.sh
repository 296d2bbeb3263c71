# Round-6 session 16: the part all-gathers through staging buffers (one contiguous collective + one strided copy per part):
# the sharded bench tests (gloo ranks sharing the device, one rank through RCCL), the one-rank line against the plain one
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s16
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests/test_gpu_bench_sharded.py -x -q > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
Q="--no-extras --no-bfs --no-refsum --no-cpu --steps 20 --warmup 5"
timeout 600 python3 bench.py $Q > $O/plain.json 2> $O/plain.log
timeout 600 python3 bench.py --force-dist --gen range $Q > $O/dist1.json 2> $O/dist1.log
timeout 900 python3 bench.py --gpus 2 --share-device --scale 24 --steps 10 --warmup 2 --no-cpu > $O/two.json 2> $O/two.log
timeout 900 python3 bench.py --scale 24 --steps 10 --warmup 2 $Q > $O/one24.json 2> $O/one24.log
python3 - <<'PY'
import json
O = "gpurun_out/r06s16"
for n in ("plain", "dist1", "two", "one24"):
    try:
        r = json.loads([l for l in open("%s/%s.json" % (O, n)) if l.startswith("{")][-1])
        print(n, "ms/step %.3f" % r["ms_per_step"], [round(x, 3) for x in r["roofline"]["kernel_ms_parts"]], r.get("rccl_ranks"), r["config"].get("parts"), "l1 %.17g" % r["pr_last_l1_change"])
    except Exception as e:
        print(n, "failed", e)
PY
