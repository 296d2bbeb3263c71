# Round-5 session 39: the default bench line twice more on one box (fresh processes): the spread of the headline and of the BFS numbers on the final code
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2; do
  timeout 900 python3 bench.py > gpurun_out/r05s39_bench$i.json 2> gpurun_out/r05s39_bench$i.log
  python3 - gpurun_out/r05s39_bench$i.json <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("PR %.3f ms frac %.3f plan %.2f s | BFS median %.3f ms = %.0f GTEPS best %.0f by source %s init %.3f | spmv %.3f tc %.2f | sssp %.3f / %.3f cc %.3f / %.3f" % (
    r["ms_per_step"], r["roofline"]["frac"], r["config"]["plan_build_s"], r["bfs"]["ms"], r["gteps_bfs"], r["gteps_bfs_best"],
    {k: round(min(v), 3) for k, v in r["bfs"]["ms_by_source"].items()}, r["bfs"]["init_ms_inside_solve"], r["spmv"]["ms"]["median"], r["tc"]["ms"]["median"],
    r["traversal"]["sssp_unit"]["ms"]["median"], r["traversal"]["sssp_u1_255_delta16"]["ms"]["median"],
    r["traversal"]["cc_with_reverse_graph"]["ms"]["median"], r["traversal"]["cc_out_edges_only"]["ms"]["median"]))
PY
done
