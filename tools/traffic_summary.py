#!/usr/bin/env python3
"""gpurun_out/<tag>/<workload>/ (tools/traffic.sh) -> profiles/<workload>_traffic.json: HBM bytes per solve and per kernel.
   python3 tools/traffic_summary.py <tag> <workload> <scale>
FETCH_SIZE is in KB and counts half of the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md "HBM"): doubled;
WRITE_SIZE in KB as it is.  Per solve = (totals at 6 solves - totals at 2 solves) / 4, kernel by kernel."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, w, scale = sys.argv[1], sys.argv[2], int(sys.argv[3])
src = os.path.join(ROOT, "gpurun_out", tag, w)


def totals(counter, reps):
    agg = collections.defaultdict(lambda: [0.0, 0])
    found = glob.glob(os.path.join(src, "%s_%d" % (counter, reps), "**", "*_counter_collection.csv"), recursive=True)
    # gpurun MERGES every session's files into gpurun_out/: one pass = one process = one file, the newest is this session's
    # (summing all of them counted a workload once per session that had profiled it)
    for f in sorted(found, key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][0] += float(r["Counter_Value"]) * 1024.0
            agg[k][1] += 1
    return agg


lo, hi = 2, 6
kern = {}
tot = 0.0
for k in sorted(set(totals("FETCH_SIZE", hi)) | set(totals("WRITE_SIZE", hi))):
    f = (totals("FETCH_SIZE", hi)[k][0] - totals("FETCH_SIZE", lo)[k][0]) / (hi - lo)
    wr = (totals("WRITE_SIZE", hi)[k][0] - totals("WRITE_SIZE", lo)[k][0]) / (hi - lo)
    n = (totals("FETCH_SIZE", hi)[k][1] - totals("FETCH_SIZE", lo)[k][1]) / (hi - lo)
    if n <= 0 or (f <= 0 and wr <= 0):
        continue  # a kernel of the graph / plan build: the same in both runs
    # kernels that run behind the solve's timed region (the statistics a solver computes for its gdn_stats: BFS's sum of the
    # reached vertices' out-degrees) are listed but not charged to the solve -- round 4 charged them: BFS 8.4 GB, of which
    # 1.6 GB were this pass over depth + rowptr
    outside = k.startswith("bfs_reached_edges") or k.startswith("gdn_reached_edges")
    kern[k] = {"dispatches_per_solve": n, "FETCH_SIZE_bytes_raw": f, "fetch_bytes_corrected_x2": 2 * f, "WRITE_SIZE_bytes": wr, "hbm_bytes": 2 * f + wr,
               "in_timed_region": not outside}
    if not outside:
        tot += 2 * f + wr
plain = open(os.path.join(src, "plain.log")).read().strip().splitlines()[-1]
ms = [float(x) for x in re.findall(r"'([0-9.]+)'", plain)]
session = " / ".join(" ".join(x.split()) for x in open(os.path.join(src, "session.txt")).read().splitlines() if x.strip())
res = {"workload": w, "scale": scale, "session": session, "hbm_bytes_per_solve": tot,
       "solve_ms_unprofiled_same_session": ms, "kernels": dict(sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes"])),
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, at 2 and 6 solves of one process each (tools/traffic.sh): "
               "per solve = the difference / 4, so graph and plan builds cancel; FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md, "
               "WRITE_SIZE (KB) as is", "line": plain}
json.dump(res, open(os.path.join(ROOT, "profiles", "%s_traffic.json" % w), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
for k, v in list(res["kernels"].items())[:12]:
    print("  %-48s x%-6.1f %10.1f MB" % (k[:48], v["dispatches_per_solve"], v["hbm_bytes"] / 1e6))
