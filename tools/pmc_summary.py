#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/profile_r01.sh into profiles/:
  * <tag>_kernel_stats.{csv,md}   from --kernel-trace --stats
  * pr_traffic.json               HBM bytes per PageRank iteration from the FETCH_SIZE / WRITE_SIZE
                                  passes (KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md
                                  "HBM": it reports 1/2 of the bytes of wide coalesced reads)
bench.py copies hbm_bytes_per_launch into roofline.traffic when scale/n_gpus match."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03_pb"
out = os.path.join(ROOT, "profiles")
SRC = "gpurun_out/" + tag.split("_")[0]  # written by tools/profile_<round>.sh (one session, one box)


def newest(pat):
    """gpurun merges every session's files into the same local directory: take the file of the LAST session"""
    return max(glob.glob(os.path.join(ROOT, pat)), key=os.path.getmtime)


def counters(pat):
    f = newest(pat)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return agg


stats = newest(os.path.join(SRC, "trace/*/*_kernel_stats.csv"))
session = " / ".join(" ".join(x.split()) for x in open(os.path.join(ROOT, SRC, "session.txt")).read().splitlines()
                     if x.strip() and not x.startswith("="))
shutil.copyfile(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(out, f"{tag}_kernel_stats.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras  (RMAT-27, PB layout)\n\n")
    f.write("Session (one box, tools/profile_%s.sh)" % tag.split("_")[0] + ": %s.  The unprofiled bench line of the same session: `profiles/%s_bench_same_session.json`.\n" % (session, tag))
    try:
        pl = json.loads([l for l in open(os.path.join(ROOT, SRC, "bench.json")) if l.startswith("{")][-1])
        ur = json.loads([l for l in open(os.path.join(ROOT, SRC, "bench_under_rocprof.json")) if l.startswith("{")][-1])
        f.write("HIP-event kernel time per iteration: unprofiled %.4f ms (A %.4f + B %.4f), under rocprofv3 %.4f ms; ms_per_step %.4f / %.4f.\n\n" % (
            pl["roofline"]["kernel_ms"], pl["roofline"]["kernel_ms_parts"][0], pl["roofline"]["kernel_ms_parts"][1], ur["roofline"]["kernel_ms"],
            pl["ms_per_step"], ur["ms_per_step"]))
    except Exception as e:  # noqa
        f.write("\n")
    f.write("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|\n")
    for r in rows[:22]:
        f.write(f"| `{r['Name'].split('(')[0][:80]}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.3f} | "
                f"{float(r['AverageNs'])/1e6:.4f} | {r['Percentage']} |\n")
if not glob.glob(os.path.join(ROOT, SRC, "fetch/*/*_counter_collection.csv")):
    # no PMC passes in this session (tools/profile_r05.sh without PMC=1): the statistics and the bench lines only
    for name in ("bench_under_rocprof.json", "bench.json", "session.txt"):
        dst = {"bench.json": f"{tag}_bench_same_session.json", "bench_under_rocprof.json": f"{tag}_bench_under_rocprof.json",
               "session.txt": f"{tag}_session.txt"}[name]
        shutil.copyfile(os.path.join(ROOT, SRC, name), os.path.join(out, dst))
    print(open(os.path.join(out, f"{tag}_kernel_stats.md")).read())
    sys.exit(0)
fetch = counters(SRC + "/fetch/*/*_counter_collection.csv")
write = counters(SRC + "/write/*/*_counter_collection.csv")
kern = {}
total = 0.0
def pick(agg, name):
    """the iterations proper: the TAG 0 instantiation (TAG 1 = the sweeps of the plan's placement search, gdn_pb.hpp)"""
    ks = [k for k in agg if name in k]
    assert len(ks) == 1, (name, list(agg))
    return agg[ks[0]]


for k, name in (("pb_expand_kernel<0>", "pb_expand_kernel<0>"), ("pb_accumulate_kernel<PrOp, 0>", "pb_accumulate_kernel<PrOp, 0>")):
    fk = sum(pick(fetch, name)) / len(pick(fetch, name)) * 1024.0
    wk = sum(pick(write, name)) / len(pick(write, name)) * 1024.0
    kern[k] = {"FETCH_SIZE_bytes_raw": fk, "fetch_bytes_corrected_x2": 2 * fk, "WRITE_SIZE_bytes": wk,
               "hbm_bytes": 2 * fk + wk, "dispatches": len(pick(fetch, name))}
    total += 2 * fk + wk
bench = json.load(open(os.path.join(ROOT, SRC, "bench_under_rocprof.json")))
plain = json.load(open(os.path.join(ROOT, SRC, "bench.json")))
res = {"scale": 27, "n_gpus": 1, "session": session, "layout": bench["config"]["layout"], "hbm_bytes_per_launch": total,
       "unprofiled_same_session": {"ms_per_step": plain["ms_per_step"], "kernel_ms": plain["roofline"]["kernel_ms"],
                                   "frac": plain["roofline"]["frac"], "step_ms": plain.get("step_ms")},
       "note": "one launch = one PageRank iteration = pb_expand_kernel<0> + pb_accumulate_kernel<PrOp, 0> (the <1> / <PrOp, 1> "
               "rows of the kernel statistics are the sweeps of the plan's placement search, not iterations); "
               "FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md, WRITE_SIZE (KB) as is; separate --pmc passes",
       "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"], "kernels": kern,
       "kernel_ms_under_rocprof": bench["roofline"]["kernel_ms"]}
json.dump(res, open(os.path.join(out, "pr_traffic.json"), "w"), indent=1)
shutil.copyfile(os.path.join(ROOT, SRC, "bench_under_rocprof.json"), os.path.join(out, f"{tag}_bench_under_rocprof.json"))
shutil.copyfile(os.path.join(ROOT, SRC, "bench.json"), os.path.join(out, f"{tag}_bench_same_session.json"))
shutil.copyfile(os.path.join(ROOT, SRC, "session.txt"), os.path.join(out, f"{tag}_session.txt"))
print(json.dumps(res, indent=1))
print(open(os.path.join(out, f"{tag}_kernel_stats.md")).read())
