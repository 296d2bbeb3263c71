#!/usr/bin/env python3
"""gpurun_out/<...>/shard_compute.json (tools/shard_compute.py) -> profiles/r06_shard_compute.md: the measured per-shard pull of
config 5 (RMAT-27 over N GPUs) next to the xGMI model of its exchange.  usage: shard_compute_md.py <json> [<out.md>]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r06_shard_compute.md")
d = json.load(open(src))
sh = d["shards"]
one = [s for s in sh if s["n"] == 1]
t1 = one[0]["ticketed"]["kernel_ms"] if one else None
with open(out, "w") as f:
    f.write("# Config 5 on the one GPU there is: the COMPUTE side of every shard, measured; the exchange, modelled\n\n")
    f.write("`python3 tools/shard_compute.py --n 1,2,4,8` (RMAT-%d, %d vertices, %d edges, %d timed pulls per line; source: `%s`).\n"
            "For every N the tool builds the shard of rank 0, of rank N/2 and of the rank with the most edges exactly as a rank of `bench.py --gpus N`\n"
            "does (`gdn_rmat_build_range` -> `gdn_pr_squish_range` -> `gdn_graph_pad_columns` -> PB plan over the padded vertex space) and times that\n"
            "shard's pull alone: phase A (`pb_expand_kernel`) / phase B (`pb_accumulate_kernel`) per launch from HIP events on the launch stream, as ONE\n"
            "launch pair (`gdn_pr_pull_dev`) and in the ticketed form the sharded driver uses (`gdn_pr_pull_parts_dev`: one launch pair, the bins in\n"
            "part order, a ticket per finished bin, one waiter kernel per part on a side stream). `frac` = the shard's own SURVEY 8(d) bytes over the\n"
            "ticketed kernel time against 8 TB/s. The replaced model table is `profiles/DESIGN_r03_history.md:658-668`.\n\n" % (
                d["scale"], d["vertices"], d["edges"], d["steps"], os.path.relpath(src, ROOT)))
    f.write("| N | rank | edges of the shard | bins = workgroups of phase B (per CU) | parts | one launch pair: A + B ms | ticketed: A + B ms (wall per pull) | frac | shard build / plan s |\n|---|---|---|---|---|---|---|---|---|\n")
    for s in sh:
        f.write("| %d | %d%s | %d (imbalance over the ranks %.1f %%) | %d (%.2f) | %d | %.3f + %.3f = %.3f | %.3f + %.3f = %.3f (%.3f) | %.3f | %.2f / %.2f |\n" % (
            s["n"], s["rank"], " (heaviest)" if s["rank"] == s["heaviest_rank"] else "", s["edges"], 100 * s["edge_imbalance"], s["bins"], s["workgroups_per_cu"],
            s["parts"], s["whole_launch"]["phase_a_ms"], s["whole_launch"]["phase_b_ms"], s["whole_launch"]["kernel_ms"],
            s["ticketed"]["phase_a_ms"], s["ticketed"]["phase_b_ms"], s["ticketed"]["kernel_ms"], s["ticketed"]["wall_ms"], s["frac_of_peak"],
            s["shard_build_s"], s["plan_build_s"]))
    f.write("\n**Compute-side scaling** (slowest measured rank of every N against the N = 1 line of this table):\n\n| N | slowest shard's kernel ms | speed-up of the compute side | efficiency |\n|---|---|---|---|\n")
    for n in sorted({s["n"] for s in sh}):
        worst = max(s["ticketed"]["kernel_ms"] for s in sh if s["n"] == n)
        if t1:
            f.write("| %d | %.3f | %.2f x | %.0f %% |\n" % (n, worst, t1 / worst, 100 * t1 / worst / n))
    f.write("\n**What does not scale, from the phases:** phase A reads the WHOLE contribution vector of the padded vertex space (252 MB at RMAT-27: every\n"
            "source chunk's slice goes through LDS whatever share of its out-edges the shard holds) -- a fixed ~0.05 ms of every shard's phase A; phase B\n"
            "runs 3-7 rounds of one-workgroup-per-CU bins whose sizes follow R-MAT's skew, so its last round is the heaviest bins' tail.\n\n")
    f.write("**The exchange (MODEL, not measured -- this box has one device; `tools/xgmi_probe.hip` and `tools/first_multi_gpu_lease.sh` are what measures it on the\n"
            "first lease with two):** a rank receives its N - 1 peers' slices of the next contribution vector (`received_bytes_per_iteration`), each over the\n"
            "pair's own xGMI link in parallel; the slice is cut into `parts` row ranges, part j's exchange is queued behind part j's tickets (phase A + (j + 1) /\n"
            "parts of phase B) and the parts queue on the link. Predicted step = when the last part has arrived.\n\n")
    keys = sorted(sh[0]["xgmi_model"].keys()) if sh else []
    f.write("| N | rank | bytes received per iteration | " + " | ".join("%s: exchange ms -> step ms -> whole-job G edges/s" % k for k in keys) + " |\n|---|---|---|" + "---|" * len(keys) + "\n")
    for s in sh:
        if s["n"] == 1:
            continue
        f.write("| %d | %d | %.1f MB | " % (s["n"], s["rank"], s["received_bytes_per_iteration"] / 1e6) +
                " | ".join("%.3f -> %.3f -> %.0f" % (s["xgmi_model"][k]["exchange_ms"], s["xgmi_model"][k]["predicted_step_ms"],
                                                     s["xgmi_model"][k]["predicted_edges_per_s_whole_job"] / 1e9) for k in keys) + " |\n")
    f.write("\nReading: from N = 2 on the step is the EXCHANGE, not the pull -- 126 MB over the one link of a pair at N = 2, 31.5 MB per peer at N = 8 against 0.65 ms of\n"
            "compute -- which is why the driver pipelines it in parts and why the bench line reports `scaling: strong` with the imbalance of the edge counts.\n")
print("wrote", out)
