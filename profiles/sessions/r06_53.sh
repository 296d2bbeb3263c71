# Round-6 session 53: why the hash-set kernel at 96 registers (dynamic LDS, five waves per SIMD asked for) makes the count 2 ms faster: each kernel alone and beside, dynamic LDS at the default bound as the control
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s53
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
for v in base dyn4 dyn5; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  for mode in beside alone; do
    spec="GDN_TC_CORE_WGS=4"; [ $mode = alone ] && spec="GDN_TC_CORE_ASYNC=0"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v.$mode -- python3 tools/tc_knob_ab.py 23 5 "$spec" > $O/$v.$mode.txt 2>&1
  done
done
python3 - <<'PY'
import glob, csv
O = "gpurun_out/r06s53"
for v in ("base", "dyn4", "dyn5"):
    for mode in ("beside", "alone"):
        line = [l for l in open("%s/%s.%s.txt" % (O, v, mode)) if "count median" in l]
        out = "%-5s %-6s %s" % (v, mode, line[-1].split("]")[1].split(" G dag")[0].strip() if line else "failed")
        for f in glob.glob("%s/%s.%s/*/*_kernel_stats.csv" % (O, v, mode)):
            for r in csv.DictReader(open(f)):
                if r["Name"].startswith("tc_count") or "tc_core_count" in r["Name"]:
                    out += " | %s avg %.3f ms (%s)" % (r["Name"].split("(")[0][-22:], float(r["AverageNs"]) / 1e6, r["Calls"])
        print(out)
PY
