#!/usr/bin/env python3
"""The COMPUTE side of config 5 (PageRank on RMAT-27, vertex-range partitioned over N GPUs) measured on the one GPU there is:

   python3 tools/shard_compute.py [--scale 27] [--n 2,4,8] [--steps 20] [--out gpurun_out/shard_compute.json]

For every N the tool builds -- exactly as rank r of `bench.py --gpus N` does (gdn_rmat_build_range, gdn_pr_squish_range,
gdn_graph_pad_columns, a PB plan over the padded vertex space) -- the shard of rank 0, of rank N/2 and of the rank with the
most edges, and times that shard's pull ALONE: the whole-iteration launch pair (gdn_pr_pull_dev) and the ticketed form the
sharded driver uses (gdn_pr_pull_parts_dev + the part waiters on a side stream), phase A / phase B per launch from HIP events,
bins, workgroups per CU, the fraction of the 8 TB/s peak on the shard's own algorithmic bytes.  What the ranks would exchange
is NOT measured (one device): the output carries an xGMI MODEL next to the measured compute (link rates are parameters), and
the predicted step of the pipeline -- part j's exchange starts when its tickets are in, exchanges queue on the links.
The degree vectors every rank would get from the two all-reduces come from one whole-graph build in front (freed again)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gardenia_amd import _cabi, graphio
from gardenia_amd.sharded import HipPageRankBackend, padded_chunk

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=27)
ap.add_argument("--edge-factor", type=int, default=16)
ap.add_argument("--n", default="2,4,8")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--link-gbs", default="48,64", help="xGMI model: effective GB/s per direction of one link, values to tabulate")
ap.add_argument("--out", default="gpurun_out/shard_compute.json")
args = ap.parse_args()

L = _cabi.lib()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
_cabi.check(L.gdn_set_device(0))
HBM = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ---- the degree vectors of the whole graph (what the ranks' all-reduces produce)
t0 = time.time()
g_out, g_in = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(args.scale, args.edge_factor, graphio.K_RAND_SEED, 1, C.byref(g_out), C.byref(g_in)))
mm, nnz_ = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(g_in, C.byref(mm), C.byref(nnz_), None, None))
m, nnz = mm.value, nnz_.value
out_degree = torch.empty(m, dtype=torch.int32, device=dev)
in_degree = torch.empty(m, dtype=torch.int32, device=dev)
_cabi.check(L.gdn_graph_degrees_dev(g_out, C.c_void_p(out_degree.data_ptr()), None))
_cabi.check(L.gdn_graph_degrees_dev(g_in, C.c_void_p(in_degree.data_ptr()), None))
torch.cuda.synchronize()
L.gdn_graph_free(g_out)
L.gdn_graph_free(g_in)
_cabi.check(L.gdn_dev_trim(None))
log(f"[shard] RMAT-{args.scale}: |V| {m} |E| {nnz}, degree vectors in {time.time() - t0:.1f} s")
in_cum = torch.cumsum(in_degree.to(torch.int64), 0)


def time_pulls(be, steps, ticketed_parts, chunk_rows=0):
    """(A ms, B ms) per launch from the plan's HIP events: `steps` pulls, ping-pong between the two contrib buffers."""
    ranges = None
    if ticketed_parts:  # (ShardedPageRank.part_ranges: equal row counts, multiples of 4, over the padded slot)
        seg = ((-(-chunk_rows // ticketed_parts)) + 3) & ~3
        ranges = [(min(j * seg, chunk_rows), chunk_rows if j == ticketed_parts - 1 else min((j + 1) * seg, chunk_rows))
                  for j in range(ticketed_parts)]
    for w in range(3 + steps):
        if w == 3:
            torch.cuda.synchronize()
            be.arm_kernel_timing(steps)
            t_wall = time.perf_counter()
        cin, cout = w & 1, (w + 1) & 1
        if ranges:
            be.pull_ticketed(cin, cout, 0.85, [r[1] for r in ranges])
            for j in range(len(ranges)):
                with be.part_ready(j):
                    pass
            torch.cuda.current_stream().wait_stream(be._side)
        else:
            be.pull(cin, cout, 0.85)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_wall) * 1e3 / steps
    (a, b), n = be.read_kernel_timing()
    be.check()
    return a / max(n, 1), b / max(n, 1), wall


results = []
for N in [int(x) for x in args.n.split(",")]:
    raw_bounds = [r * m // N for r in range(N + 1)]
    edges = [int((in_cum[raw_bounds[r + 1] - 1] - (in_cum[raw_bounds[r] - 1] if raw_bounds[r] else 0)).item()) for r in range(N)]
    heaviest = max(range(N), key=lambda r: edges[r])
    ranks = []
    for r in (0, N // 2, heaviest):
        if r not in ranks:
            ranks.append(r)
    for rank in ranks:
        v_lo, v_hi = raw_bounds[rank], raw_bounds[rank + 1]
        t0 = time.time()
        scratch_out = torch.zeros(m, dtype=torch.int32, device=dev)  # (the rank's own out-degree part: not needed here)
        rows = C.c_void_p()
        _cabi.check(L.gdn_rmat_build_range(args.scale, args.edge_factor << args.scale, 0.57, 0.19, 0.19, graphio.K_RAND_SEED, 1,
                                           v_lo, v_hi, C.byref(rows), C.c_void_p(scratch_out.data_ptr())))
        del scratch_out
        rb, sb = (C.c_int32 * (N + 1))(*raw_bounds), (C.c_int32 * (N + 1))()
        _cabi.check(L.gdn_pr_squish_range(rows, v_lo, C.c_void_p(in_degree.data_ptr()), C.c_void_p(out_degree.data_ptr()), m, N + 1, rb, sb))
        bl = list(sb)
        chunk = padded_chunk(bl)
        _cabi.check(L.gdn_graph_pad_columns(rows, N, sb, chunk))
        live = (in_degree[v_lo:v_hi] > 0) | (out_degree[v_lo:v_hi] > 0)
        deg_local = out_degree[v_lo:v_hi][live].contiguous()
        del live
        sm, snnz = C.c_int32(), C.c_uint64()
        _cabi.check(L.gdn_graph_info(rows, C.byref(sm), C.byref(snnz), None, None))
        m_space, lo = chunk * N, rank * chunk
        hi = lo + (bl[rank + 1] - bl[rank])
        torch.cuda.synchronize()
        t_build = time.time() - t0
        t0 = time.time()
        be = HipPageRankBackend(torch, rows, deg_local, m_space, lo, hi, chunk, N, dev, layout=-1, m_base=m, force_sharded=True)
        torch.cuda.synchronize()
        t_plan = time.time() - t0
        for c in be.contribs:  # every slot a valid contribution (values do not move the timing; they must lie in [0, 1])
            c.fill_(1.0 / m / 16.0)
        be.contrib(0)
        nb = be.n_bins()
        parts = max(1, min(4, nb // 128))  # (bench.py's rule)
        a1, b1, w1 = time_pulls(be, args.steps, 0)
        a4, b4, w4 = time_pulls(be, args.steps, parts, chunk)
        ib = be.iter_bytes()
        rec = {"n": N, "rank": rank, "heaviest_rank": heaviest, "edges": snnz.value, "edges_per_rank": edges,
               "edge_imbalance": max(edges) * N / sum(edges) - 1.0, "rows": sm.value, "chunk": chunk, "bins": nb,
               "log_blk": be.log_blk, "workgroups_per_cu": nb / 256.0, "parts": parts,
               "whole_launch": {"phase_a_ms": a1, "phase_b_ms": b1, "kernel_ms": a1 + b1, "wall_ms": w1},
               "ticketed": {"phase_a_ms": a4, "phase_b_ms": b4, "kernel_ms": a4 + b4, "wall_ms": w4},
               "algorithmic_bytes": ib, "frac_of_peak": ib / ((a4 + b4) * 1e-3) / 1e9 / HBM,
               "received_bytes_per_iteration": 4 * chunk * (N - 1), "shard_build_s": t_build, "plan_build_s": t_plan}
        # ---- xGMI model (NOT measured: one device).  Fully connected node, one link per peer pair: a rank receives its
        # (N - 1) peers' slices in parallel, each over its own link; part j's exchange starts when part j is final (phase A +
        # (j + 1) / parts of phase B, the parts being ranges of equal row counts) and the parts queue on the links.
        rec["xgmi_model"] = {}
        for gbs in [float(x) for x in args.link_gbs.split(",")]:
            tx = 4.0 * chunk / (gbs * 1e9) * 1e3  # ms: one peer's slice over one link
            fin = 0.0
            for j in range(parts):
                ready = a4 + b4 * (j + 1) / parts
                fin = max(ready, fin) + tx / parts
            rec["xgmi_model"]["%g GB/s per link" % gbs] = {"exchange_ms": tx, "predicted_step_ms": fin,
                                                          "predicted_edges_per_s_whole_job": nnz / (fin * 1e-3)}
        log(f"[shard] {json.dumps(rec)}")
        results.append(rec)
        be.close()
        del be, deg_local
        L.gdn_graph_free(rows)
        torch.cuda.empty_cache()
        _cabi.check(L.gdn_dev_trim(None))

os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
json.dump({"scale": args.scale, "vertices": m, "edges": nnz, "steps": args.steps, "shards": results}, open(args.out, "w"), indent=1)
print(json.dumps({"shards": len(results), "out": args.out}))
