# Round-6 session 3: (a) the new reference-order sums (gdn_seqsum.hpp, grouped launches): parity tests, then their price on the
# headline; (b) tickets with write-through stores against the release fence; (c) shard bins spread over whole rounds (N = 8)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s03
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "reference or ticketed" > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 1500 python3 -m pytest tests/test_gpu_configs.py -x -q -k "reference or converge or lj" > $O/pytest_b.txt 2>&1; tail -3 $O/pytest_b.txt
Q="--no-extras --no-bfs --no-cpu --steps 20 --warmup 5"
GDN_PR_SUM_TRACE=1 timeout 900 python3 bench.py $Q > $O/plain.json 2> $O/plain.log; grep "pr refsum" $O/plain.log | head -3
GDN_PR_SUM_GROUP_LOG=19 timeout 900 python3 bench.py $Q > $O/plain_g19.json 2> $O/plain_g19.log
GDN_PR_SUM_GROUP_LOG=21 timeout 900 python3 bench.py $Q --refsum-min-degree 50000 > $O/plain_g21_d50k.json 2> $O/plain_g21_d50k.log
timeout 900 python3 bench.py $Q --refsum-min-degree 2000 > $O/plain_d2k.json 2> $O/plain_d2k.log
timeout 600 python3 bench.py --force-dist --gen range $Q > $O/dist1_wt.json 2> $O/dist1_wt.log
GDN_PR_TICKET_MODE=fence timeout 600 python3 bench.py --force-dist --gen range $Q > $O/dist1_fence.json 2> $O/dist1_fence.log
timeout 1500 python3 tools/shard_compute.py --n 1,8 --out $O/shard_compute.json > $O/shard.out 2> $O/shard.log; tail -2 $O/shard.log | cut -c1-300
python3 - <<'PY'
import json
O = "gpurun_out/r06s03"
for n in ("plain", "plain_g19", "plain_g21_d50k", "plain_d2k", "dist1_wt", "dist1_fence"):
    try:
        r = json.loads([l for l in open("%s/%s.json" % (O, n)) if l.startswith("{")][-1])
        rs = r.get("pr_reference_sum") or {}
        print(n, "ms/step %.3f" % r["ms_per_step"], "kernel_ms %.3f" % r["roofline"]["kernel_ms"], [round(x, 3) for x in r["roofline"]["kernel_ms_parts"]], "plan %.2f s" % r["config"]["plan_build_s"],
              "| refsum ms %.3f rows %s longest %s entries %s plan %.2f l1 %s vs %s" % (rs.get("ms_per_step", 0), rs.get("rows_resummed"), rs.get("longest_row"), rs.get("entries_resummed"), rs.get("plan_build_s", 0), rs.get("pr_last_l1_change"), r["pr_last_l1_change"]))
    except Exception as e:
        print(n, "failed:", e)
try:
    for s in json.load(open(O + "/shard_compute.json"))["shards"]:
        print("N %d rank %d edges %d bins %d wg/cu %.2f parts %d whole A %.3f B %.3f | ticketed A %.3f B %.3f wall %.3f | frac %.3f" % (
            s["n"], s["rank"], s["edges"], s["bins"], s["workgroups_per_cu"], s["parts"], s["whole_launch"]["phase_a_ms"], s["whole_launch"]["phase_b_ms"],
            s["ticketed"]["phase_a_ms"], s["ticketed"]["phase_b_ms"], s["ticketed"]["wall_ms"], s["frac_of_peak"]))
except Exception as e:
    print("shard_compute failed:", e)
PY
