#!/usr/bin/env python3
"""Per-configuration table of the counter passes of tools/pr_channels.sh: for each pass the dispatches of the two iteration
kernels in dispatch order, 4 per configuration (the first is the untimed one and is dropped), durations from the kernel trace
of the same process.  usage: pr_channels_summary.py <outdir> <configs>"""
import collections
import csv
import glob
import sys

out, configs = sys.argv[1], int(sys.argv[2])
KERNELS = ("pb_expand_kernel<0>", "pb_accumulate_kernel<PrOp, 0>")
for d in sorted(glob.glob(out + "/p[0-9]*/")):
    cc = glob.glob(d + "**/*_counter_collection.csv", recursive=True)
    kt = glob.glob(d + "**/*_kernel_trace.csv", recursive=True)
    if not cc or not kt:
        print(d, "no output")
        continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    per = {k: collections.OrderedDict() for k in KERNELS}  # kernel -> dispatch id -> {counter: value}
    for r in csv.DictReader(open(cc[0])):
        for k in KERNELS:
            if r["Kernel_Name"].startswith(k) or (k in r["Kernel_Name"]):
                per[k].setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    print("==", d, open(d.rstrip("/") + ".log").read().count("config"), "configurations in the log")
    for k in KERNELS:
        ids = list(per[k])
        if len(ids) < 4 * configs:
            print("  %s: %d dispatches (expected %d)" % (k, len(ids), 4 * configs))
            continue
        ids = ids[-4 * configs:]  # (the plan build runs none of these two kernels; be safe anyway)
        names = sorted(per[k][ids[0]])
        print("  %-32s %8s  %s" % (k[:32], "ms", "  ".join("%14s" % n[-14:] for n in names)))
        for c in range(configs):
            grp = ids[4 * c + 1: 4 * c + 4]
            ms = sum(dur[i] for i in grp) / len(grp)
            vals = [sum(per[k][i][n] for i in grp) / len(grp) for n in names]
            print("  config %-25d %8.3f  %s" % (c, ms, "  ".join("%14.5g" % v for v in vals)))
