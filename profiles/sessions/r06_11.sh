# Round-6 session 11: (a) reference-order sums after the last changes (WG rows capped at 256, a round of memory ahead, two streams,
# groups of 2^22): tests + price; (b) PageRank at config 2's size (LJ-like): slice geometry knobs on the experiments build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s11
mkdir -p $O; rm -rf $O/*
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "reference or ticketed" > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
Q="--no-extras --no-bfs --no-cpu --steps 20 --warmup 5"
timeout 900 python3 bench.py $Q > $O/plain.json 2> $O/plain.log
GDN_PR_SUM_GROUP_LOG=21 timeout 900 python3 bench.py $Q > $O/plain_g21.json 2> $O/plain_g21.log
GDN_PR_SUM_GROUP_LOG=23 timeout 900 python3 bench.py $Q > $O/plain_g23.json 2> $O/plain_g23.log
timeout 900 python3 bench.py $Q --refsum-min-degree 50000 > $O/plain_d50k.json 2> $O/plain_d50k.log
export GDN_PR_SUM=reference GDN_PR_SUM_MIN_DEGREE=10000
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ref -- python3 bench.py $Q --no-refsum > $O/refsum_rocprof.json 2> $O/trace_ref.log
unset GDN_PR_SUM GDN_PR_SUM_MIN_DEGREE
python3 - <<'PY'
import json, glob, csv
O = "gpurun_out/r06s11"
for n in ("plain", "plain_g21", "plain_g23", "plain_d50k", "refsum_rocprof"):
    try:
        r = json.loads([l for l in open("%s/%s.json" % (O, n)) if l.startswith("{")][-1])
        rs = r.get("pr_reference_sum") or {}
        print(n, "ms/step %.3f" % r["ms_per_step"], "| refsum ms %.3f rows %s entries %s launches %s l1 %s vs %s" % (rs.get("ms_per_step", 0), rs.get("rows_resummed"), rs.get("entries_resummed"), rs.get("launches_per_iteration_for_the_resum"), rs.get("pr_last_l1_change"), r["pr_last_l1_change"]))
    except Exception as e:
        print(n, "failed:", e)
for f in glob.glob("%s/trace_ref/*/*_kernel_stats.csv" % O):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("  %-70s calls %5s total %9.3f ms avg %8.4f ms" % (r["Name"].split("(")[0][:70], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
export GARDENIA_HIP_LIB=$GRAFT_REPO_ROOT/gardenia_amd/lib/var_exp/libgardenia_hip.so GDN_TEST_HOOKS=1
timeout 900 python3 tools/pr_midsize.py "" "GDN_PB_BALANCE_ALL=1" "GDN_PB_SLICES_LOG=10" "GDN_PB_SLICES_LOG=10,GDN_PB_BALANCE_ALL=1" "GDN_PB_SLICES_LOG=8" "GDN_PB_LOG_CHUNK=13,GDN_PB_LOG_BIN=13" "GDN_PB_LOG_CHUNK=15,GDN_PB_LOG_BIN=14" "GDN_PB_LOG_CHUNK=15,GDN_PB_LOG_BIN=13" > $O/midsize.txt 2>&1; cat $O/midsize.txt | tail -10
