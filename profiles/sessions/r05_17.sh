# Round-5 session 17: the profile session (tools/profile_r05.sh: unprofiled bench line, rocprofv3 --kernel-trace --stats of the same command, bench again)
bash tools/profile_r05.sh > gpurun_out/r05_profile.log 2>&1; tail -5 gpurun_out/r05_profile.log
python3 - <<'PY'
import json
for f in ("bench.json", "bench_under_rocprof.json", "bench_after.json"):
    try:
        d = json.loads([l for l in open("gpurun_out/r05/" + f) if l.startswith("{")][-1])
        print(f, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_parts"], d["config"]["plan_build_s"], d.get("gteps_bfs"))
    except Exception as e:
        print(f, "failed", e)
PY
