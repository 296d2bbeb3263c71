# Round-6 session 8: where does a launch of pr_refseg_kernel spend its 60-80 us?  Timing-only ablations on the experiments build
# (gardenia_amd/lib/var_exp, -DGDN_EXPERIMENTS): GDN_PR_REF_DBG 1 = no scan, 2 = no gather, 3 = neither; 379 rows (>= 50 000 in-edges)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s08
mkdir -p $O; rm -rf $O/*
export GARDENIA_HIP_LIB=$GRAFT_REPO_ROOT/gardenia_amd/lib/var_exp/libgardenia_hip.so
Q="--no-extras --no-bfs --no-cpu --steps 10 --warmup 3 --no-refsum"
export GDN_PR_SUM=reference GDN_PR_SUM_MIN_DEGREE=50000
for dbg in 0 1 2 3; do
  GDN_PR_REF_DBG=$dbg timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$dbg -- python3 bench.py $Q > $O/dbg_$dbg.json 2> $O/trace_$dbg.log
done
export GDN_PR_SUM_MIN_DEGREE=10000
for dbg in 0 1 2 3; do
  GDN_PR_REF_DBG=$dbg timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace10k_$dbg -- python3 bench.py $Q > $O/dbg10k_$dbg.json 2> $O/trace10k_$dbg.log
done
python3 - <<'PY'
import glob, csv
O = "gpurun_out/r06s08"
for t in ("trace_0", "trace_1", "trace_2", "trace_3", "trace10k_0", "trace10k_1", "trace10k_2", "trace10k_3"):
    for f in glob.glob("%s/%s/*/*_kernel_stats.csv" % (O, t)):
        for r in csv.DictReader(open(f)):
            if "pr_refseg" in r["Name"] or "pr_ref_apply" in r["Name"]:
                print("%-12s %-28s calls %5s avg %8.4f ms" % (t, r["Name"].split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
