#!/usr/bin/env python3
"""Re-attach profiles/<workload>_traffic.json to a committed bench line: every `roofline` object that carries a
`traffic_source` gets `traffic` / `frac_traffic` recomputed from the CURRENT json of its workload and the solve time its old
pair implies (time = traffic / frac_traffic / peak).  Needed once: tools/traffic_summary.py summed the counter files of
every session gpurun had merged into gpurun_out/ (x4 by the end of round 4) and the bench line of the last session was
printed before that was found.   python3 tools/reattach_traffic.py profiles/r04_pb_bench_same_session.json"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 8000.0e9
path = sys.argv[1]
lines = open(path).read().strip().splitlines()
d = json.loads(lines[-1])


def walk(o):
    if isinstance(o, dict):
        src = o.get("traffic_source")
        if isinstance(src, str) and o.get("traffic") and o.get("frac_traffic"):
            w = re.match(r"profiles/(\w+)_traffic\.json", src)
            if w and w.group(1) != "pr":  # (pr_traffic.json comes from tools/pmc_summary.py, which takes the newest files)
                tj = json.load(open(os.path.join(ROOT, "profiles", w.group(1) + "_traffic.json")))
                seconds = o["traffic"] / o["frac_traffic"] / PEAK
                o["traffic"] = tj["hbm_bytes_per_solve"]
                o["frac_traffic"] = tj["hbm_bytes_per_solve"] / seconds / PEAK
                o["traffic_source"] = "profiles/%s_traffic.json (%s), re-attached by tools/reattach_traffic.py" % (w.group(1), tj.get("session", ""))
                print("%-10s %.3f ms  traffic %.2f GB  frac_traffic %.3f" % (w.group(1), seconds * 1e3, o["traffic"] / 1e9, o["frac_traffic"]))
        for v in o.values():
            walk(v)
    elif isinstance(o, list):
        for v in o:
            walk(v)


walk(d)
lines[-1] = json.dumps(d)
open(path, "w").write("\n".join(lines) + "\n")
