# Round-6 session 30: the chunk stream's look-ups without lane masks (TcSet::count_fast): counts (tests), time beside the core and alone; the two-instruction hash again
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s30
mkdir -p $O; rm -rf $O/*
export GDN_TEST_HOOKS=1
timeout 1200 python3 -m pytest tests -x -q -m gpu -k "tc or triangle" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in base hxor; do
  if [ $v = base ]; then unset GARDENIA_HIP_LIB; else export GARDENIA_HIP_LIB=$PWD/gardenia_amd/lib/var_$v/libgardenia_hip.so; fi
  for mode in beside alone; do
    spec=""; [ $mode = alone ] && spec="GDN_TC_CORE_ASYNC=0"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v.$mode -- python3 tools/tc_knob_ab.py 23 5 "$spec" > $O/$v.$mode.txt 2>&1
  done
  timeout 600 python3 tools/tc_knob_ab.py orkut 6 "" > $O/orkut_$v.txt 2>&1; tail -2 $O/orkut_$v.txt | head -1
  timeout 600 python3 tools/tc_knob_ab.py 21 6 "" > $O/r21_$v.txt 2>&1; tail -2 $O/r21_$v.txt | head -1
done
python3 - <<'PY'
import glob, csv
O = "gpurun_out/r06s30"
for v in ("base", "hxor"):
    for mode in ("beside", "alone"):
        line = [l for l in open("%s/%s.%s.txt" % (O, v, mode)) if "count median" in l]
        out = "%-7s %-6s %s" % (v, mode, line[-1].split("]")[1].split(" G dag")[0].strip() if line else "failed")
        for f in glob.glob("%s/%s.%s/*/*_kernel_stats.csv" % (O, v, mode)):
            for r in csv.DictReader(open(f)):
                if r["Name"].startswith("tc_count") or "tc_core_count" in r["Name"]:
                    out += " | %s avg %.3f ms (%s)" % (r["Name"].split("(")[0][-22:], float(r["AverageNs"]) / 1e6, r["Calls"])
        print(out)
PY
