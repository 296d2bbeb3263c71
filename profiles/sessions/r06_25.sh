# Round-6 session 25: the hash-set kernel's compile-time knobs again, now that the work counters no longer bound it (chunks in flight, set size / waves per SIMD, slice)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s25
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
V="base unr8 unr2 cap512w8 cap512w6 slice1024 slice256"
for v in $V; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  for mode in beside alone; do
    spec=""; [ $mode = alone ] && spec="GDN_TC_CORE_ASYNC=0"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v.$mode -- python3 tools/tc_knob_ab.py 23 5 "$spec" > $O/$v.$mode.txt 2>&1
  done
done
V="$V" python3 - <<'PY'
import glob, csv, os
O = "gpurun_out/r06s25"
for v in os.environ["V"].split():
    for mode in ("beside", "alone"):
        line = [l for l in open("%s/%s.%s.txt" % (O, v, mode)) if "count median" in l]
        out = "%-9s %-6s %s" % (v, mode, line[-1].split("]")[1].split(" G dag")[0].strip() if line else "failed")
        for f in glob.glob("%s/%s.%s/*/*_kernel_stats.csv" % (O, v, mode)):
            for r in csv.DictReader(open(f)):
                if r["Name"].startswith("tc_count") or "tc_core_count" in r["Name"]:
                    out += " | %s avg %.3f ms (%s)" % (r["Name"].split("(")[0][-22:], float(r["AverageNs"]) / 1e6, r["Calls"])
        print(out)
PY
