#!/usr/bin/env python3
"""gpurun_out/r06 (tools/profile_r06.sh: one box, one session) -> profiles/r06_pb_kernel_stats.{csv,md}, r06_pb_bench_same_session.json,
r06_pb_bench_under_rocprof.json and profiles/pr_traffic.json (HBM bytes per PageRank iteration: FETCH_SIZE x 2 + WRITE_SIZE of the two
kernels' launches of the iterations proper, KB units, MI355X_MICROARCH.md "HBM").  bench.py copies hbm_bytes_per_launch into roofline.traffic."""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06")
out = os.path.join(ROOT, "profiles")
A, B = "pb_expand_kernel<0>", "pb_accumulate_kernel<PrOp, 0>"


def newest(pat):
    return max(glob.glob(os.path.join(SRC, pat), recursive=True), key=os.path.getmtime)


session = " / ".join(" ".join(x.split()) for x in open(os.path.join(SRC, "session.txt")).read().splitlines() if x.strip())
stats = newest("trace/**/*_kernel_stats.csv")
shutil.copyfile(stats, os.path.join(out, "r06_pb_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
pl = json.loads([l for l in open(os.path.join(SRC, "bench.json")) if l.startswith("{")][-1])
ur = json.loads([l for l in open(os.path.join(SRC, "bench_under_rocprof.json")) if l.startswith("{")][-1])
ka = {k: next((float(r["AverageNs"]) / 1e6 for r in rows if k in r["Name"]), 0.0) for k in (A, B)}
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(c + "/**/*_counter_collection.csv"))):
        for k in (A, B):
            if k in r["Kernel_Name"]:
                agg[k].append(float(r["Counter_Value"]))
    per[c] = {k: sum(v) / max(len(v), 1) for k, v in agg.items()}
byt = {k: (2 * per["FETCH_SIZE"].get(k, 0) + per["WRITE_SIZE"].get(k, 0)) * 1024 for k in (A, B)}
total = byt[A] + byt[B]
with open(os.path.join(out, "r06_pb_kernel_stats.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-refsum  (RMAT-27, PB layout, final code of round 6)\n\n")
    f.write("Session (one box, tools/profile_r06.sh): %s.  The same session holds the unprofiled bench line (`profiles/r06_pb_bench_same_session.json`) and the FETCH_SIZE / WRITE_SIZE passes behind `profiles/pr_traffic.json`.\n" % session)
    f.write("HIP-event kernel time per iteration: unprofiled %.4f ms (A %.4f + B %.4f), under rocprofv3 %.4f ms; ms_per_step %.4f / %.4f.\n" % (
        pl["roofline"]["kernel_ms"], pl["roofline"]["kernel_ms_parts"][0], pl["roofline"]["kernel_ms_parts"][1], ur["roofline"]["kernel_ms"], pl["ms_per_step"], ur["ms_per_step"]))
    f.write("Kernel statistics: A %.4f + B %.4f = **%.4f ms** -> %.0f GB/s on the %.3f GB of SURVEY 8(d) = **%.3f** of 8 TB/s.  Counter traffic: A %.3f + B %.3f = %.3f GB per iteration = %.3f x algorithmic.\n\n" % (
        ka[A], ka[B], ka[A] + ka[B], pl["roofline"]["algorithmic_bytes_per_launch"] / ((ka[A] + ka[B]) * 1e-3) / 1e9, pl["roofline"]["algorithmic_bytes_per_launch"] / 1e9,
        pl["roofline"]["algorithmic_bytes_per_launch"] / ((ka[A] + ka[B]) * 1e-3) / 1e9 / 8000.0, byt[A] / 1e9, byt[B] / 1e9, total / 1e9, total / pl["roofline"]["algorithmic_bytes_per_launch"]))
    f.write("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|\n")
    for r in rows[:22]:
        f.write("| `%s` | %s | %.3f | %.4f | %s |\n" % (r["Name"].split("(")[0][:80], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
for name, dst in (("bench.json", "r06_pb_bench_same_session.json"), ("bench_under_rocprof.json", "r06_pb_bench_under_rocprof.json"), ("session.txt", "r06_pb_session.txt")):
    shutil.copyfile(os.path.join(SRC, name), os.path.join(out, dst))
json.dump({"scale": 27, "n_gpus": 1, "hbm_bytes_per_launch": total, "expand_bytes": byt[A], "accumulate_bytes": byt[B],
           "fetch_kb": per["FETCH_SIZE"], "write_kb": per["WRITE_SIZE"], "session": "round 6, tools/profile_r06.sh: " + session,
           "note": "FETCH_SIZE x 2 + WRITE_SIZE (KB) per launch of pb_expand_kernel<0> + pb_accumulate_kernel<PrOp, 0>, separate --pmc passes"},
          open(os.path.join(out, "pr_traffic.json"), "w"), indent=1)
print("kernel stats %.4f + %.4f = %.4f ms; traffic %.3f GB" % (ka[A], ka[B], ka[A] + ka[B], total / 1e9))
