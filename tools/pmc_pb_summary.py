#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/ (tools/pmc_pb.sh) -> profiles/<tag>_phaseB_counters.md: per-launch averages of every counter for
pb_expand_kernel<0> (A) and pb_accumulate_kernel<PrOp, 0> (B), the derived occupancies of round 3's table
(profiles/r03_pb_phaseB_counters.md) and the kernel statistics of the unprofiled trace of the same session."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "pmc_" + tag)
A, B = "pb_expand_kernel<0>", "pb_accumulate_kernel<PrOp, 0>"
val = {A: {}, B: {}}
for f in sorted(glob.glob(src + "/p*/**/*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for k in (A, B):
            if k in r["Kernel_Name"]:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        val[k][c] = (sum(v) / len(v), len(v))
stats = {}
for f in glob.glob(src + "/trace/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in (A, B):
            if k in r["Name"]:
                stats[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
out = os.path.join(ROOT, "profiles", tag + "_phaseB_counters.md")
with open(out, "w") as f:
    sess = " / ".join(x.strip() for x in open(src + "/session.txt").read().splitlines() if x.strip()) if os.path.exists(src + "/session.txt") else ""
    f.write("# Hardware counters of the two kernels of the PageRank iteration (RMAT-27, squished PB plan, final code of round 6)\n\n")
    f.write("`tools/pmc_pb.sh %s` on one box (%s): one `rocprofv3 --pmc <set> --kernel-trace --kernel-include-regex` pass per counter set over\n"
            "`tools/attic/pr_notorch.py 27 2`; per-launch averages over the launches of the iterations proper (the `<1>` instances of the placement\n"
            "search are left out). Chip totals; `GRBM_GUI_ACTIVE` is summed over the 8 XCDs.\n\n" % (tag, sess))
    if stats:
        f.write("Kernel statistics of the same session (`--kernel-trace --stats`, no counters): A %.4f ms x %d, B %.4f ms x %d.\n\n" % (
            stats.get(A, (0, 0))[1], stats.get(A, (0, 0))[0], stats.get(B, (0, 0))[1], stats.get(B, (0, 0))[0]))
    f.write("| counter | `pb_expand_kernel<0>` (A) | `pb_accumulate_kernel<PrOp, 0>` (B) | launches |\n|---|---|---|---|\n")
    for c in sorted(set(val[A]) | set(val[B])):
        f.write("| `%s` | %.4g | %.4g | %d |\n" % (c, val[A].get(c, (0, 0))[0], val[B].get(c, (0, 0))[0], val[B].get(c, (0, 0))[1]))
    f.write("\nDerived (256 CUs, 1024 SIMDs, one texture-address unit and one LDS per CU):\n\n| quantity | A | B |\n|---|---|---|\n")

    def g(k, c):
        return val[k].get(c, (0.0, 0))[0]

    rows = []
    cyc = {k: g(k, "GRBM_GUI_ACTIVE") / 8 for k in (A, B)}
    rows.append(("kernel cycles (GRBM / 8)", "%.3g" % cyc[A], "%.3g" % cyc[B]))
    for name, fn in (
        ("VALU issue: `SQ_INSTS_VALU` x 4 cycles / 1024 SIMDs / kernel cycles", lambda k: g(k, "SQ_INSTS_VALU") * 4 / 1024 / max(cyc[k], 1)),
        ("LDS busy: `SQ_LDS_IDX_ACTIVE` / 256 / kernel cycles", lambda k: g(k, "SQ_LDS_IDX_ACTIVE") / 256 / max(cyc[k], 1)),
        ("of it bank conflicts", lambda k: g(k, "SQ_LDS_BANK_CONFLICT") / max(g(k, "SQ_LDS_IDX_ACTIVE"), 1)),
        ("texture-address unit busy: `TA_TA_BUSY_sum` / 256 / kernel cycles", lambda k: g(k, "TA_TA_BUSY_sum") / 256 / max(cyc[k], 1)),
        ("vector L1 stalled on pending misses: `TCP_PENDING_STALL_CYCLES_sum` / 256 / kernel cycles", lambda k: g(k, "TCP_PENDING_STALL_CYCLES_sum") / 256 / max(cyc[k], 1)),
        ("L2 hit rate `TCC_HIT / (HIT + MISS)`", lambda k: g(k, "TCC_HIT_sum") / max(g(k, "TCC_HIT_sum") + g(k, "TCC_MISS_sum"), 1)),
        ("waves waiting: `SQ_WAIT_ANY` / `SQ_WAVE_CYCLES`", lambda k: g(k, "SQ_WAIT_ANY") / max(g(k, "SQ_WAVE_CYCLES"), 1)),
    ):
        rows.append((name, "%.0f %%" % (100 * fn(A)), "%.0f %%" % (100 * fn(B))))
    rows.append(("TA cycles per vector-memory instruction", "%.0f" % (g(A, "TA_TA_BUSY_sum") / max(g(A, "SQ_INSTS_VMEM_RD"), 1)),
                 "%.0f" % (g(B, "TA_TA_BUSY_sum") / max(g(B, "SQ_INSTS_VMEM_RD"), 1))))
    rows.append(("HBM bytes per launch: FETCH_SIZE x 2 + WRITE_SIZE (KB units; MI355X_MICROARCH.md, HBM)",
                 "%.3f GB" % ((g(A, "FETCH_SIZE") * 2 + g(A, "WRITE_SIZE")) * 1024 / 1e9), "%.3f GB" % ((g(B, "FETCH_SIZE") * 2 + g(B, "WRITE_SIZE")) * 1024 / 1e9)))
    for r in rows:
        f.write("| %s | %s | %s |\n" % r)
print("wrote", out)
