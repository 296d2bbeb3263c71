# usage: bash tools/prof_stats.sh <name> <python script + args ...>   -> gpurun_out/prof_<name>/ (kernel stats csv)
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python3 "$@" > gpurun_out/prof_$name.out 2> gpurun_out/prof_$name.log
f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-70s %8s %12s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
for r in rows[:25]:
    print("%-70s %8s %12.3f %10.1f %6.1f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
