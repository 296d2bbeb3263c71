#!/usr/bin/env python3
"""bench.py -- PageRank pull iterations on a synthetic R-MAT graph (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--scale 27] [--edge-factor 16]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE PageRank pull iteration (gather + score update + L1 norm + next contrib; the
work of contrib/pull_step/l1norm in src/pr/base.cu:115-121) over the whole graph, inputs
resident in HBM.  At N=1 the workload is RMAT scale 27, avg degree 16 (north star / config 5),
generated on the device by gdn_rmat_build.  N>1 shards the SAME graph by vertex range (strong
scaling) with one RCCL all-gather of the contrib vector per step.

Rank 0 prints ONE JSON line: metric value = whole-job edges/s; "roofline" = algorithmic bytes
of one iteration (8(m+1)+4nnz+4nnz+16m) / per-launch duration of the dominant kernel
(HIP events on the launch stream) against the 8 TB/s HBM peak; "cpu_baseline" = the CPU oracle's
OpenMP pull iteration on a bounded row sample of the same graph; extra fields report BFS GTEPS
on the same graph (BFS stays single GPU).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=int, default=27)
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--layout", choices=["auto", "csr", "pb"], default="auto",
                    help="edge layout of the PageRank plan (include/gardenia_hip.h GDN_LAYOUT_*)")
    ap.add_argument("--share-device", action="store_true",
                    help="TEST ONLY: every rank uses cuda:0 and the collectives go through gloo, so the N>1 code path "
                         "can be exercised on a 1-GPU box (numbers are meaningless)")
    ap.add_argument("--exchange", choices=["auto", "dense", "compact"], default="auto",
                    help="N > 1: contributions exchanged per iteration -- every row (dense) or only the rows with "
                         "out-edges (compact = auto)")
    ap.add_argument("--no-squish", action="store_true",
                    help="keep the vertices without any edge in the per-iteration state (the caller's vertex space)")
    ap.add_argument("--no-bfs", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the baseline sample")
    args = ap.parse_args()

    import numpy as np
    import torch  # before libgardenia_hip: both must share ONE libamdhip64 (same SONAME)
    import torch.distributed as dist
    from gardenia_amd import _cabi, graphio
    from gardenia_amd.sharded import HipPageRankBackend, ShardedPageRank, vertex_range

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: gardenia_amd has no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    L = _cabi.lib()
    _cabi.check(L.gdn_set_device(local_rank))
    if world > 1:
        if args.share_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    # ---- synthetic input: R-MAT(scale, edge_factor), cleaned like the reference loader
    t0 = time.time()
    g_out, g_in = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build(args.scale, args.edge_factor, graphio.K_RAND_SEED, 1, C.byref(g_out), C.byref(g_in)))
    m, nnz = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(g_in, C.byref(m), C.byref(nnz), None, None))
    m, nnz = m.value, nnz.value
    out_degree = torch.empty(m, dtype=torch.int32, device=device)
    _cabi.check(L.gdn_graph_degrees_dev(g_out, C.c_void_p(out_degree.data_ptr()), None))
    torch.cuda.synchronize()
    t_build = time.time() - t0
    if rank == 0:
        log(f"[bench] RMAT-{args.scale} x{args.edge_factor}: |V| {m} |E| {nnz} built on device in {t_build:.1f} s")

    # ---- this rank's shard.  N > 1: the graph is first relabelled to its live vertices (gdn_pr_squish_*: the vertices
    # without any edge keep the base score and are left out of the per-iteration state), then cut into vertex ranges of
    # that space; N = 1: the plan does the same internally (GDN_LAYOUT_PB_SQUISHED)
    if args.no_squish:
        os.environ["GDN_PR_SQUISH"] = "0"
    squish_first = world > 1 and args.layout != "csr" and os.environ.get("GDN_PR_SQUISH", "1") != "0"
    m_part, g_part, deg_part, m_base, sq = m, g_in, out_degree, 0, None
    if squish_first:
        sq = C.c_void_p()
        _cabi.check(L.gdn_pr_squish_create(g_in, C.c_void_p(out_degree.data_ptr()), C.byref(sq)))
        ms_, gp = C.c_int32(0), C.c_void_p()
        _cabi.check(L.gdn_pr_squish_info(sq, None, C.byref(ms_), C.byref(gp), None))
        m_part, g_part, m_base = ms_.value, gp, m
        deg_part = torch.empty(m_part, dtype=torch.int32, device=device)
        _cabi.check(L.gdn_pr_squish_degrees_dev(sq, C.c_void_p(deg_part.data_ptr()), None))
        torch.cuda.synchronize()
    lo, hi, chunk = vertex_range(rank, world, m_part)
    shard = g_part
    if world > 1:
        shard = C.c_void_p()
        _cabi.check(L.gdn_graph_slice_rows(g_part, lo, hi, C.byref(shard)))
    sm, snnz = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(shard, C.byref(sm), C.byref(snnz), None, None))
    deg_local = deg_part[lo:hi].contiguous()
    t0 = time.time()
    be = HipPageRankBackend(torch, shard, deg_local, m_part, lo, hi, chunk, world, device,
                            layout={"auto": -1, "csr": 0, "pb": 1}[args.layout], m_base=m_base)
    torch.cuda.synchronize()
    t_plan = time.time() - t0
    layout_name = {0: "natural vertex order, in-CSR u64 offsets / i32 ids, merge-path tiles",
                   1: "propagation-blocked tiles (source chunk of %d ids x destination bin of %d rows, u16 local "
                      "ids, LDS-resident slices, 2^-62 fixed-point LDS accumulation)"
                      % (1 << (be.log_blk // 100), 1 << (be.log_blk % 100))}[be.layout]
    if squish_first:
        layout_name += "; vertex space of the %d live vertices (%.1f %% of |V|: the others have no edge and keep the base " \
                       "score), relabelled before the vertex-range cut" % (m_part, 100.0 * m_part / m)
    if be.layout == 1 and be.squished:
        layout_name += "; vertex space of the %d live vertices (%.1f %% of |V|: the others have no edge, keep the base " \
                       "score and are written once at export)" % (be.m_state, 100.0 * be.m_state / m)
    if be.layout == 1:
        nh, he = C.c_int32(0), C.c_uint64(0)
        _cabi.check(L.gdn_pr_plan_hubs(be.plan, C.byref(nh), C.byref(he)))
        mt, ms_, me = C.c_int32(0), C.c_int32(0), C.c_uint64(0)
        _cabi.check(L.gdn_pr_plan_mid(be.plan, C.byref(mt), C.byref(ms_), C.byref(me)))
        if nh.value or mt.value:
            layout_name += "; record tiers: the %d hub sources (%.1f %% of this rank's edges) and %d mid-tier sources " \
                           "(%.1f %%) bypass the expand phase -- the accumulate phase reads their edges as 32-bit " \
                           "(source, row) records sorted by source, values from per-iteration tables of fixed-point codes" \
                           % (nh.value, 100.0 * he.value / max(snnz.value, 1), ms_.value,
                              100.0 * me.value / max(snnz.value, 1))
    # on a squished graph 82 % of the state has out-edges: the dense in-place all-gather beats gather + scatter
    exchange = "dense" if (squish_first and args.exchange == "auto") else args.exchange
    # pipeline parts: every part's accumulate launch should still fill the 256 CUs (a part of 104 bins at N = 8 would
    # leave 60 % of them idle): about 200 bins per part or more, the same count on every rank
    parts = 4
    if world > 1:
        nb = torch.tensor([be.n_bins()], dtype=torch.int64, device=device)
        dist.all_reduce(nb, op=dist.ReduceOp.MIN)
        parts = max(1, min(4, int(nb.item()) // 200)) if int(nb.item()) > 0 else 4
    pr = ShardedPageRank(be, m_part, rank, world, dist if world > 1 else None, exchange=exchange, parts=parts)
    pr.init_contrib()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pr.step()
    barrier()
    be.arm_kernel_timing(args.steps)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        pr.step()
    barrier()
    elapsed = time.perf_counter() - t_start
    (kA_ms, kB_ms), klaunches = be.read_kernel_timing()
    ktot_ms = kA_ms + kB_ms
    last_err = pr.global_diff()
    be.check()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        kt = torch.tensor([ktot_ms / max(klaunches, 1)], dtype=torch.float64, device=device)
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
        k_avg_ms = float(kt.item())
    else:
        k_avg_ms = ktot_ms / max(klaunches, 1)
    ms_per_step = elapsed * 1e3 / args.steps
    value = nnz * args.steps / elapsed  # whole-job edges per second

    # roofline of the dominant kernel (merge-path tile kernel), per launch, on this rank's shard
    iter_bytes = be.iter_bytes()
    achieved = iter_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
    traffic = None
    tf = os.path.join(ROOT, "profiles", "pr_traffic.json")
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            if tj.get("scale") == args.scale and tj.get("n_gpus") == world:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "PR-iter edges/s (pull PageRank, RMAT-%d avg-deg %d)" % (args.scale, args.edge_factor),
        "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "PageRank pull iteration on R-MAT scale %d, avg degree %d (Graph500 "
                               "A=.57 B=.19 C=.19, seed 27491095, self loops+duplicates dropped)"
                               % (args.scale, args.edge_factor),
                   "vertices": m, "edges": nnz, "layout": layout_name, "plan_build_s": t_plan,
                   "partition": "vertex-range x%d, RCCL all-gather of contrib (%s exchange, %.0f MB received per rank "
                                "and iteration) pipelined in %d row-range parts behind the pull kernels"
                                % (world, pr.exchange, pr.exchanged_bytes() / 1e6, pr.parts) if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "mp_tile_kernel<PrOp>" if be.layout == 0 else
                     "pb_expand_kernel + pb_accumulate_kernel<PrOp> (one iteration = both)",
                     "kernel_ms": k_avg_ms, "launches": klaunches,
                     "kernel_ms_parts": [kA_ms / max(klaunches, 1), kB_ms / max(klaunches, 1)],
                     "algorithmic_bytes_per_launch": iter_bytes},
        "gteps_pr": value / 1e9, "pr_last_l1_change": last_err, "graph_build_s": t_build,
    }

    # ---- BFS GTEPS on the same graph (single GPU: the N = 1 run carries it; BFS stays single-GPU per north_star, and the
    # other ranks of an N > 1 job would only wait in the final barrier for it), outside the timed region
    if rank == 0 and world == 1 and not args.no_bfs:
        try:
            # source: first vertices with out-degree > 0
            nz = torch.nonzero(out_degree[:1 << 16] > 0)[:4].flatten().tolist()
            dist_buf = torch.empty(m, dtype=torch.int32, device=device)
            best = None
            t0 = time.time()
            bplan = C.c_void_p()
            _cabi.check(L.gdn_bfs_plan_create(g_out, g_in, 1, C.byref(bplan)))
            t_bplan = time.time() - t0
            for s in nz[:3]:
                st = _cabi.GdnStats()
                _cabi.check(L.gdn_bfs_run(bplan, int(s), C.c_void_p(dist_buf.data_ptr()), C.byref(st)))
                gteps = st.edges_traversed / (st.solve_ms * 1e-3) / 1e9 if st.solve_ms > 0 else 0.0
                rec = {"source": int(s), "ms": st.solve_ms, "levels": st.iterations,
                       "edges_traversed": st.edges_traversed, "gteps": gteps}
                if st.edges_traversed > nnz // 100 and (best is None or gteps > best["gteps"]):
                    best = rec
                log(f"[bench] BFS from {s}: {rec}")
            L.gdn_bfs_plan_free(bplan)
            if best:
                best["plan_build_s"] = t_bplan
                out["bfs"] = best
                out["gteps_bfs"] = best["gteps"]
        except Exception as e:  # BFS is an extra; never lose the PR line
            log(f"[bench] BFS skipped: {e}")

    # ---- CPU baseline: the oracle's OpenMP pull iteration on a bounded row sample (rank 0, N=1)
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            from oracle import binding as orc
            t1 = time.time()
            h_rp = np.empty(m + 1, np.uint64)
            h_ci = np.empty(nnz, np.int32)
            _cabi.check(L.gdn_graph_download(g_in, h_rp.ctypes.data_as(C.c_void_p), h_ci.ctypes.data_as(C.c_void_p)))
            h_deg = out_degree.cpu().numpy()
            gi = graphio.CSR(m, h_rp, h_ci)
            scores = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
            cores = orc.num_threads()
            # probe 1/64 of the rows to size the sample
            probe_hi = max(1, m // 64)
            tp = time.time()
            orc.pr_iterate(gi, h_deg, scores, 1, row_lo=0, row_hi=probe_hi)
            tp = time.time() - tp
            frac = min(1.0, max(1.0 / 64, (args.cpu_seconds / max(tp, 1e-3)) / 64))
            row_hi = max(1, int(m * frac))
            tc = time.time()
            orc.pr_iterate(gi, h_deg, scores, 1, row_lo=0, row_hi=row_hi)
            tc = time.time() - tc
            e_sample = int(h_rp[row_hi])
            out["cpu_baseline"] = {"value": e_sample / tc, "unit": "edges/s", "cores": cores, "kind": "port",
                                   "sample": "1 pull iteration over rows [0,%d) of the same RMAT-%d graph "
                                             "(%d edges, %.1f%% of the graph) incl. the contrib pass over all "
                                             "vertices, OpenMP restatement of src/pr/omp_base.cc:23-34"
                                             % (row_hi, args.scale, e_sample, 100.0 * e_sample / nnz),
                                   "seconds": tc}
            log(f"[bench] cpu baseline: {out['cpu_baseline']} (download+prep {time.time() - t1 - tc:.1f} s)")
        except Exception as e:
            log(f"[bench] cpu baseline skipped: {e}")
            out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out), flush=True)
    be.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
