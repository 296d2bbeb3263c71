# Round-6 session 7: BFS, the slow source: searches in issue order (which run jumps?), per-level traces,
# and the counter traffic of the slow source (tools/traffic.sh bfs:1 -- the second non-isolated source, 5 on RMAT-27)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s07
mkdir -p $O; rm -rf $O/*
timeout 900 python3 tools/bfs_runs.py 27 6 2 > $O/runs.txt 2>&1; grep -E "^round|traced" $O/runs.txt
for reps in 2 6; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "bfs_|gdn_|GdnMailbox|mailbox" --output-format csv -d $O/${c}_$reps -- python3 tools/traffic_run.py bfs:1 27 $reps > $O/${c}_$reps.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
O = "gpurun_out/r06s07"
tot = {}
for reps in (2, 6):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(float)
        for f in glob.glob("%s/%s_%d/**/*_counter_collection.csv" % (O, c, reps), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"].split("(")[0][:48]] += float(r["Counter_Value"])
        tot[(c, reps)] = agg
kernels = sorted(set(tot[("FETCH_SIZE", 6)]) | set(tot[("WRITE_SIZE", 6)]))
print("bytes per search (difference of 6 and 2 searches / 4; FETCH_SIZE x 2 + WRITE_SIZE, KB units):")
total = 0.0
for k in kernels:
    f = (tot[("FETCH_SIZE", 6)].get(k, 0) - tot[("FETCH_SIZE", 2)].get(k, 0)) / 4
    w = (tot[("WRITE_SIZE", 6)].get(k, 0) - tot[("WRITE_SIZE", 2)].get(k, 0)) / 4
    b = (2 * f + w) * 1024
    if b > 1e6:
        print("  %-50s %8.3f GB" % (k, b / 1e9))
    if "reached_edges" not in k:
        total += b
print("  total (without the statistics pass behind the timed region): %.3f GB" % (total / 1e9))
PY
