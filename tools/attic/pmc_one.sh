# usage: bash tools/pmc_one.sh "<counters>" <kernel-substring> <python script + args...>   (one PMC pass)
set=$1; kern=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_one; mkdir -p gpurun_out/pmc_one
rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_one -- python3 "$@" > gpurun_out/pmc_one.log 2>&1
python3 - "$kern" <<'PY'
import csv, glob, collections, sys
kern = sys.argv[1]
for f in glob.glob("gpurun_out/pmc_one/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        if kern in k[0]:
            print("%-42s %-32s n=%3d avg=%.4g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
