# Round-6 session 20: where does a round of pr_refscan_wg_kernel spend its 6.5 us?  Timing-only ablations (experiments build):
# GDN_PR_REF_DBG 1 = no pairs, 2 = no chain, 3 = neither; rows of >= 50 000 in-edges (256 workgroup rows + 123 wave rows)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s20
mkdir -p $O; rm -rf $O/*
export GARDENIA_HIP_LIB=$GRAFT_REPO_ROOT/gardenia_amd/lib/var_exp/libgardenia_hip.so
Q="--no-extras --no-bfs --no-cpu --steps 10 --warmup 3 --no-refsum"
export GDN_PR_SUM=reference GDN_PR_SUM_MIN_DEGREE=50000
for dbg in 0 1 2 3; do
  GDN_PR_REF_DBG=$dbg timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$dbg -- python3 bench.py $Q > $O/dbg_$dbg.json 2> $O/trace_$dbg.log
done
python3 - <<'PY'
import glob, csv
O = "gpurun_out/r06s20"
for t in ("trace_0", "trace_1", "trace_2", "trace_3"):
    for f in glob.glob("%s/%s/*/*_kernel_stats.csv" % (O, t)):
        for r in csv.DictReader(open(f)):
            if "pr_ref" in r["Name"] and int(r["Calls"]) > 5:
                print("%-9s %-26s calls %5s avg %8.4f ms" % (t, r["Name"].split("(")[0][:26], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
