# Round-6 session 10: reference-order sums -- is the grouping needed at all?  glog 31 (one launch over whole rows) / 24 / 22 with
# the rows of >= 125 000 in-edges on workgroups (GDN_PR_SUM_WG_MIN, a test hook), against the default (glog 20)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s10
mkdir -p $O; rm -rf $O/*
Q="--no-extras --no-bfs --no-cpu --steps 20 --warmup 5"
export GDN_TEST_HOOKS=1 GDN_PR_SUM_WG_MIN=125000
for gl in 22 21 23 20; do
  GDN_PR_SUM_GROUP_LOG=$gl timeout 900 python3 bench.py $Q > $O/g$gl.json 2> $O/g$gl.log
done
GDN_PR_SUM_WG_MIN=60000 GDN_PR_SUM_GROUP_LOG=31 timeout 900 python3 bench.py $Q > $O/g31_wg60k.json 2> $O/g31_wg60k.log
GDN_PR_SUM_WG_MIN=250000 GDN_PR_SUM_GROUP_LOG=31 timeout 900 python3 bench.py $Q > $O/g31_wg250k.json 2> $O/g31_wg250k.log
GDN_PR_SUM_GROUP_LOG=31 timeout 900 python3 bench.py $Q --refsum-min-degree 50000 > $O/g31_d50k.json 2> $O/g31_d50k.log
GDN_PR_SUM_GROUP_LOG=31 timeout 900 python3 bench.py $Q --refsum-min-degree 2000 > $O/g31_d2k.json 2> $O/g31_d2k.log
python3 - <<'PY'
import json
O = "gpurun_out/r06s10"
for n in ("g31", "g24", "g22", "g20", "g31_wg60k", "g31_wg250k", "g31_d50k", "g31_d2k"):
    try:
        r = json.loads([l for l in open("%s/%s.json" % (O, n)) if l.startswith("{")][-1])
        rs = r.get("pr_reference_sum") or {}
        print(n, "ms/step %.3f" % r["ms_per_step"], "| refsum ms %.3f rows %s entries %s launches %s" % (rs.get("ms_per_step", 0), rs.get("rows_resummed"), rs.get("entries_resummed"), rs.get("launches_per_iteration_for_the_resum")))
    except Exception as e:
        print(n, "failed:", e)
PY
