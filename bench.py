#!/usr/bin/env python3
"""bench.py -- PageRank pull iterations on a synthetic R-MAT graph (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--scale 27] [--edge-factor 16]

N > 1 works either way: launched by torch.distributed.run (one rank per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
the environment), or as plain `python bench.py --gpus N` -- the script then starts the N ranks itself (a
torch.distributed.run child process, spawned BEFORE this process touches a GPU; its exit code becomes ours).

A "step" is ONE PageRank pull iteration (gather + score update + L1 norm + next contrib; the work of
contrib/pull_step/l1norm in src/pr/base.cu:115-121) over the whole graph, inputs resident in HBM.  At N=1 the workload is
RMAT scale 27, avg degree 16 (north star / config 5), generated on the device by gdn_rmat_build.  N>1 shards the SAME
graph by vertex range (strong scaling; ranges of about nnz/N edges) with one RCCL all-gather of the contrib
vector per step.

Rank 0 prints ONE JSON line: metric value = whole-job edges/s; "roofline" = algorithmic bytes of one iteration (SURVEY 8d:
8(m+1)+4nnz+4nnz+16m) / per-launch duration of the dominant kernels (HIP events on the launch stream) against the 8 TB/s
HBM peak; "cpu_baseline" = the CPU oracle's OpenMP pull iteration on a bounded row sample of the same graph.  At N=1 the
line also carries, measured OUTSIDE the PageRank timed region on the same box: "bfs" (GTEPS + bytes/roofline on the same
graph), "spmv" (BASELINE config 3: fp32 SpMV on RMAT-25, resident plan and the one-shot drop-in) and "tc" (config 4's
stand-in: triangle count on symmetrized RMAT-23, Orkut-sized) and "traversal" (SSSP unit / U[1,255] weights and CC on
RMAT-24) -- each with median + min over >= 10 repetitions.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run the N ranks as a torch.distributed.run child (fresh processes;
    this parent has not touched the GPU and never does), pass its output through, return its exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] --gpus {n} without WORLD_SIZE in the environment: starting {n} ranks ({' '.join(cmd[1:8])} ...)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env, cwd=ROOT)


def med_min(xs):
    xs = sorted(xs)
    n = len(xs)
    return {"median": (xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])), "min": xs[0], "n": n}


def attach_traffic(roof, workload, scale, seconds):
    """Counter-measured HBM bytes of one solve from profiles/<workload>_traffic.json (tools/traffic.sh: rocprofv3 FETCH_SIZE x 2 +
    WRITE_SIZE passes of tools/traffic_run.py on this workload; NOT measured in this run) -> roofline.traffic and
    roofline.frac_traffic = traffic / solve time / peak: the HBM utilisation, next to the model-based `frac`."""
    roof["traffic"], roof["frac_traffic"], roof["traffic_source"] = None, None, None
    f = os.path.join(ROOT, "profiles", "%s_traffic.json" % workload)
    try:
        tj = json.load(open(f))
        if tj.get("scale") == scale and seconds > 0:
            roof["traffic"] = tj["hbm_bytes_per_solve"]
            roof["frac_traffic"] = tj["hbm_bytes_per_solve"] / seconds / 1e9 / HBM_PEAK_GBS
            roof["traffic_source"] = "profiles/%s_traffic.json (%s), not this run" % (workload, tj.get("session", "session not recorded"))
    except (OSError, ValueError, KeyError):
        pass
    return roof


def physical_cores():
    """Physical cores of this host: distinct (physical id, core id) pairs of /proc/cpuinfo; SMT siblings count once."""
    try:
        cores, cur = set(), {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif not line.strip() and cur:
                if "core id" in cur:
                    cores.add((cur.get("physical id", "0"), cur["core id"]))
                cur = {}
        if cur and "core id" in cur:
            cores.add((cur.get("physical id", "0"), cur["core id"]))
        if cores:
            return len(cores)
    except OSError:
        pass
    return os.cpu_count() or 1


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
    ap.add_argument("--force-dist", action="store_true",
                    help="TEST ONLY: run the N>1 code path (process group, padded ranges, collectives) with the ranks "
                         "there are, even one -- a 1-GPU box then drives the RCCL backend through every call")
    ap.add_argument("--exchange", choices=["auto", "dense", "compact"], default="auto",
                    help="N > 1: contributions exchanged per iteration -- every row (dense) or only the rows with "
                         "out-edges (compact = auto)")
    ap.add_argument("--ranges", choices=["balanced", "equal"], default="balanced",
                    help="N > 1: vertex ranges of about nnz/N edges each (SURVEY 8e) or of equal vertex counts")
    ap.add_argument("--gen", choices=["auto", "whole", "range"], default="auto",
                    help="N > 1: every rank builds the whole graph and cuts its shard out (whole), or generates only the in-edges "
                         "of its own destination range and the ranks all-reduce the degree vectors (range: gdn_rmat_build_range + "
                         "gdn_pr_squish_range; no rank holds the whole graph; ranges of equal vertex counts of the permuted ids). "
                         "auto = range for N > 1 on the blocked layout with the live-vertex relabelling, else whole")
    ap.add_argument("--no-squish", action="store_true",
                    help="keep the vertices without any edge in the per-iteration state (the caller's vertex space)")
    ap.add_argument("--no-bfs", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the SpMV (RMAT-25) and TC (RMAT-23) blocks")
    ap.add_argument("--spmv-scale", type=int, default=25)
    ap.add_argument("--tc-scale", type=int, default=23)
    ap.add_argument("--trav-scale", type=int, default=24, help="R-MAT scale of the SSSP / CC block")
    ap.add_argument("--standin-shrink", type=int, default=0,
                    help="TEST ONLY: the LJ-like / Orkut-like stand-ins at 2^-k of their size (scale - k, draws >> k)")
    ap.add_argument("--reps", type=int, default=12, help="repetitions behind every median / min")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the baseline sample")
    ap.add_argument("--parts", type=int, default=0,
                    help="N > 1: row-range parts the exchange is pipelined in (0 = by the bin count of the smallest rank)")
    ap.add_argument("--reserve-gb", type=float, default=0.0,
                    help="MEASUREMENT: set this much device memory aside as the first device call of the process (gdn_dev_reserve); the "
                         "PageRank plan's per-iteration scratch array then lives there (DESIGN 4.1, placement)")
    ap.add_argument("--no-refsum", action="store_true",
                    help="skip pr_reference_sum: the same iteration under GDN_PR_SUM=reference (the rows of >= 10^4 in-edges re-summed "
                         "in the reference's fp32 order), a second plan on the same graph, outside the timed region")
    ap.add_argument("--refsum-min-degree", type=int, default=10000)
    ap.add_argument("--no-converged-parity", action="store_true",
                    help="skip parity_note.converged: the timed plan solved to epsilon 1e-4 against the CPU oracle's full solve "
                         "(about a minute of host time at RMAT-27 on 128 cores)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    # the host baseline runs on all PHYSICAL cores, OMP_PROC_BIND=spread.  Set before any OpenMP runtime is
    # loaded (torch brings one); a caller's own settings win.
    # Only the process that runs the CPU baseline takes them (rank 0 of an N = 1 job): the ranks of an N > 1 launch would
    # each bind a full-machine thread team to the same cores -- they get their share of the cores instead.
    n_phys = physical_cores()
    n_ranks_here = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
    if n_ranks_here <= 1:
        os.environ.setdefault("OMP_NUM_THREADS", str(n_phys))
        os.environ.setdefault("OMP_PROC_BIND", "spread")
        os.environ.setdefault("OMP_PLACES", "cores")
    else:
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, n_phys // n_ranks_here)))

    import numpy as np
    import torch  # before libgardenia_hip: both must share ONE libamdhip64 (same SONAME)
    import torch.distributed as dist
    from gardenia_amd import _cabi, graphio
    from gardenia_amd.sharded import HipPageRankBackend, ShardedPageRank, padded_chunk

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; the job has WORLD_SIZE ranks and reports n_gpus = {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: gardenia_amd has no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    L = _cabi.lib()
    _cabi.check(L.gdn_set_device(local_rank))
    if args.reserve_gb > 0:
        _cabi.check(L.gdn_dev_reserve(int(args.reserve_gb * (1 << 30))))
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if args.share_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    # ---- synthetic input: R-MAT(scale, edge_factor), cleaned like the reference loader
    t0 = time.time()
    if args.no_squish:
        os.environ["GDN_PR_SQUISH"] = "0"
    squish_first = multi and args.layout != "csr" and os.environ.get("GDN_PR_SQUISH", "1") != "0"
    gen_range = squish_first and (args.gen == "range" or (args.gen == "auto" and world > 1))
    if args.gen == "range" and not gen_range:
        raise SystemExit("bench.py: --gen range needs N > 1 (or --force-dist), the blocked layout and the live-vertex relabelling")
    g_out = g_in = range_rows = None
    if gen_range:
        # every rank: the in-edges of ITS destination range only (all edge indices generated, the range's keys kept), its part of
        # the out-degree count; the two degree vectors are completed by all-reduces -- no rank holds the whole graph
        m = 1 << args.scale
        if world > m:
            raise SystemExit(f"bench.py: {world} ranks for {m} vertices -- every rank needs a row")
        raw_bounds = [r * m // world for r in range(world + 1)]
        v_lo, v_hi = raw_bounds[rank], raw_bounds[rank + 1]
        out_degree = torch.zeros(m, dtype=torch.int32, device=device)
        in_degree = torch.zeros(m, dtype=torch.int32, device=device)
        range_rows = C.c_void_p()
        _cabi.check(L.gdn_rmat_build_range(args.scale, args.edge_factor << args.scale, 0.57, 0.19, 0.19, graphio.K_RAND_SEED, 1,
                                           v_lo, v_hi, C.byref(range_rows), C.c_void_p(out_degree.data_ptr())))
        _cabi.check(L.gdn_graph_degrees_dev(range_rows, C.c_void_p(in_degree[v_lo:v_hi].data_ptr()), None))
        torch.cuda.synchronize()
        if world > 1:
            dist.all_reduce(out_degree)
            dist.all_reduce(in_degree)
        nnz = int(in_degree.sum(dtype=torch.int64).item())
        args.no_bfs = args.no_extras = True  # (blocks of the N = 1 line: they need the whole graph)
        if "--ranges" in sys.argv and args.ranges == "balanced" and rank == 0:
            log("[bench] warning: --ranges balanced has no effect with per-rank generation (--gen range): every rank owns an equal "
                "vertex range of the permuted ids; the line reports the edge imbalance (config.edge_imbalance). --gen whole cuts "
                "ranges of equal edge counts.")
    else:
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
        log(f"[bench] RMAT-{args.scale} x{args.edge_factor}: |V| {m} |E| {nnz} built on device in {t_build:.1f} s"
            + (" (per rank: the in-edges of its own destination range)" if gen_range else ""))
    if world > m:
        raise SystemExit(f"bench.py: {world} ranks for {m} vertices -- every rank needs a row")  # every rank exits alike

    # ---- this rank's shard.  N > 1: the graph is first relabelled to its live vertices (gdn_pr_squish_*: the vertices
    # without any edge keep the base score and are left out of the per-iteration state), then cut into vertex ranges of
    # that space -- of about nnz/N edges each -- and moved into the padded vertex space (gdn_graph_slice_padded) in which
    # every rank's slice of the contrib vector is an equal all-gather slot; N = 1: the plan squishes internally
    # (GDN_LAYOUT_PB_SQUISHED)
    m_part, g_part, deg_part, sq = m, g_in, out_degree, None
    dead_diff = 0.0
    deg_state_host = None  # rank 0 of an N > 1 job with a CPU leg: the out-degrees in the order of the sharded vertex space
    want_cpu_multi = multi and rank == 0 and not args.no_cpu
    if gen_range:
        rb, sb = (C.c_int32 * (world + 1))(*raw_bounds), (C.c_int32 * (world + 1))()
        _cabi.check(L.gdn_pr_squish_range(range_rows, v_lo, C.c_void_p(in_degree.data_ptr()), C.c_void_p(out_degree.data_ptr()), m,
                                          world + 1, rb, sb))
        bl = list(sb)
        m_part = bl[world]
        live = (in_degree[v_lo:v_hi] > 0) | (out_degree[v_lo:v_hi] > 0)
        deg_local = out_degree[v_lo:v_hi][live].contiguous()
        if want_cpu_multi:
            deg_state_host = out_degree[(in_degree > 0) | (out_degree > 0)].cpu().numpy()
        del in_degree, live
        base, start = np.float32((np.float32(1.0) - np.float32(0.85)) / np.float32(m)), np.float32(1.0) / np.float32(m)
        dead_diff = float(m - m_part) * float(abs(np.float32(base - start)))
    elif squish_first:
        sq = C.c_void_p()
        _cabi.check(L.gdn_pr_squish_create(g_in, C.c_void_p(out_degree.data_ptr()), C.byref(sq)))
        ms_, gp = C.c_int32(0), C.c_void_p()
        _cabi.check(L.gdn_pr_squish_info(sq, None, C.byref(ms_), C.byref(gp), None))
        m_part, g_part = ms_.value, gp
        deg_part = torch.empty(m_part, dtype=torch.int32, device=device)
        _cabi.check(L.gdn_pr_squish_degrees_dev(sq, C.c_void_p(deg_part.data_ptr()), None))
        torch.cuda.synchronize()
        # the dead vertices move from 1/m to the base score in the first iteration: |base - 1/m| each
        base, start = np.float32((np.float32(1.0) - np.float32(0.85)) / np.float32(m)), np.float32(1.0) / np.float32(m)
        dead_diff = float(m - m_part) * float(abs(np.float32(base - start)))
    part_nnz = None
    if gen_range:
        chunk = padded_chunk(bl)
        _cabi.check(L.gdn_graph_pad_columns(range_rows, world, sb, chunk))
        shard = range_rows
        b_lo, b_hi = bl[rank], bl[rank + 1]
        m_space, lo, hi = chunk * world, rank * chunk, rank * chunk + (b_hi - b_lo)
    elif multi:
        bounds = (C.c_int32 * (world + 1))()
        if args.ranges == "balanced":
            _cabi.check(L.gdn_graph_balanced_ranges(g_part, world, bounds))
        else:
            per = -(-m_part // world)
            for r in range(world + 1):
                bounds[r] = min(r * per, m_part)
        bl = list(bounds)
        chunk = padded_chunk(bl)
        shard = C.c_void_p()
        _cabi.check(L.gdn_graph_slice_padded(g_part, world, bounds, chunk, rank, C.byref(shard)))
        b_lo, b_hi = bl[rank], bl[rank + 1]
        m_space, lo, hi = chunk * world, rank * chunk, rank * chunk + (b_hi - b_lo)
        deg_local = deg_part[b_lo:b_hi].contiguous()
        if want_cpu_multi:
            deg_state_host = deg_part.cpu().numpy()
    else:
        shard, chunk, m_space, lo, hi, deg_local = g_part, m_part, m_part, 0, m_part, deg_part
    sm, snnz = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(shard, C.byref(sm), C.byref(snnz), None, None))
    if multi:
        t = torch.zeros(world, dtype=torch.int64, device=device)
        t[rank] = snnz.value
        if world > 1:
            dist.all_reduce(t)
        part_nnz = [int(v) for v in t.tolist()]
    t0 = time.time()
    be = HipPageRankBackend(torch, shard, deg_local, m_space, lo, hi, chunk, world, device,
                            layout={"auto": -1, "csr": 0, "pb": 1}[args.layout], m_base=m if multi else 0,
                            force_sharded=multi)
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
    if multi:
        nb = torch.tensor([be.n_bins()], dtype=torch.int64, device=device)
        dist.all_reduce(nb, op=dist.ReduceOp.MIN)
        # (with tickets -- gdn_pr_pull_parts_dev -- the parts are ranges of ONE launch and cost no tail between them.  The bins of a
        # part run largest-first only among themselves: four parts of ONE round of workgroups each cost RMAT-27 / 8's accumulate
        # phase 7 %, 0.03 ms -- but from N = 2 on the step is the EXCHANGE (0.5-1.6 ms a slice, profiles/r06_shard_compute.md), and
        # every part that can leave early is worth a quarter of it: four parts wherever a part still holds half a round)
        parts = max(1, min(4, int(nb.item()) // 128)) if int(nb.item()) > 0 else 4
    if multi and args.parts > 0:
        parts = min(args.parts, 8)
    pr = ShardedPageRank(be, m_space, rank, world, dist if multi else None, exchange=exchange, parts=parts,
                         first_diff_extra=dead_diff, force_collectives=args.force_dist)
    pr.init_contrib()

    def barrier():
        torch.cuda.synchronize()
        if multi:
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
    # median + min of single, individually synchronised steps (>= 10 repetitions), outside the timed region
    rep_ms = []
    for _ in range(max(args.reps, 1)):
        barrier()
        t1 = time.perf_counter()
        pr.step()
        barrier()
        rep_ms.append((time.perf_counter() - t1) * 1e3)
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        kt = torch.tensor([ktot_ms / max(klaunches, 1)], dtype=torch.float64, device=device)
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
        k_avg_ms = float(kt.item())
        rt = torch.tensor(rep_ms, dtype=torch.float64, device=device)
        dist.all_reduce(rt, op=dist.ReduceOp.MAX)
        rep_ms = rt.tolist()
    else:
        k_avg_ms = ktot_ms / max(klaunches, 1)
    ms_per_step = elapsed * 1e3 / args.steps
    value = nnz * args.steps / elapsed  # whole-job edges per second

    # roofline of the dominant kernels, per launch, on this rank's shard
    iter_bytes = be.iter_bytes()
    achieved = iter_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
    # HBM traffic: NOT measured in this run (PMC counters need rocprofv3 around the process); the figure of the
    # committed counter session, with where and when it was taken, or null
    traffic, traffic_src = None, None
    tf = os.path.join(ROOT, "profiles", "pr_traffic.json")
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            if tj.get("scale") == args.scale and tj.get("n_gpus") == world and args.layout == "auto":
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_src = "profiles/pr_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command " \
                              "(%s), not this run" % tj.get("session", "session not recorded")
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
                   "partition": "vertex-range x%d (%s; edges per rank %s), RCCL all-gather of contrib (%s exchange, "
                                "%.0f MB received per rank and iteration) pipelined in %d row-range parts behind the pull "
                                "kernels" % (world, "equal ranges of the permuted ids (every rank generated its own destination range, "
                                             "the degree vectors were all-reduced)" if gen_range else args.ranges + " ranges", part_nnz,
                                             pr.exchange, pr.exchanged_bytes() / 1e6, pr.parts)
                   if multi else "single GPU"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "mp_tile_kernel<PrOp>" if be.layout == 0 else
                     "pb_expand_kernel<0> + pb_accumulate_kernel<PrOp, 0> (one iteration = both; <1> = the plan's placement search)",
                     "kernel_ms": k_avg_ms, "launches": klaunches,
                     "kernel_ms_parts": [kA_ms / max(klaunches, 1), kB_ms / max(klaunches, 1)],
                     "algorithmic_bytes_per_launch": iter_bytes,
                     # the same model charged for the vertices the plan touches only (a squished plan iterates on the
                     # live ones; the others keep the base score): 8 (m_live + 1) + 8 nnz + 16 m_live
                     "frac_live_vertices": (((8 * (be.m_state + 1) + 8 * snnz.value + 16 * be.m_state) / (k_avg_ms * 1e-3) / 1e9
                                             / HBM_PEAK_GBS) if (be.layout == 1 and getattr(be, "squished", False) and k_avg_ms > 0 and not multi)
                                            else None)},
        "step_ms": med_min(rep_ms),
        "gteps_pr": value / 1e9, "pr_last_l1_change": last_err, "graph_build_s": t_build,
    }
    if multi:
        # how many ranks the collective backend itself reports (the world the all-gathers ran over), and which backend
        out["rccl_ranks"] = int(dist.get_world_size())
        out["collective_backend"] = str(dist.get_backend()) + (" (= RCCL on ROCm)" if str(dist.get_backend()) == "nccl" else
                                                               " (--share-device test mode: not RCCL)")
        out["config"]["parts"] = pr.parts
        out["config"]["bins_per_rank_min"] = int(nb.item())
        if part_nnz:
            out["config"]["edges_per_rank"] = part_nnz
            out["config"]["edge_imbalance"] = max(part_nnz) * len(part_nnz) / max(sum(part_nnz), 1) - 1.0

    # ---- the price of the contract-exact iteration: the same plan built under GDN_PR_SUM=reference with the
    # rows of >= 10^4 in-edges re-summed in the order of src/pr/omp_base.cc:27-30 behind every pull (those rows then carry the
    # reference's bits, and no row of RMAT-27 lies beyond north_star's 1e-4: tests/test_gpu_configs.py); outside the timed region
    if rank == 0 and world == 1 and not multi and not args.no_refsum and be.layout == 1:
        try:
            _cabi.check(L.gdn_option_set(b"GDN_PR_SUM", b"reference"))
            _cabi.check(L.gdn_option_set(b"GDN_PR_SUM_MIN_DEGREE", str(args.refsum_min_degree).encode()))
            t0 = time.time()
            be2 = HipPageRankBackend(torch, shard, deg_local, m_space, lo, hi, chunk, world, device,
                                     layout={"auto": -1, "csr": 0, "pb": 1}[args.layout])
            torch.cuda.synchronize()
            t_plan2 = time.time() - t0
            pr2 = ShardedPageRank(be2, m_space, 0, 1, None, parts=1)
            pr2.init_contrib()
            for _ in range(args.warmup):
                pr2.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pr2.step()
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t1) * 1e3 / args.steps
            rows_, longest_, entries_, groups_ = C.c_int32(0), C.c_int32(0), C.c_uint64(0), C.c_int32(0)
            _cabi.check(L.gdn_pr_plan_refsum_info(be2.plan, C.byref(rows_), C.byref(longest_), C.byref(entries_), C.byref(groups_)))
            be2.check()
            out["pr_reference_sum"] = {
                "ms_per_step": ms2, "frac": iter_bytes / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "default_ms_per_step": ms_per_step,
                "min_in_degree": args.refsum_min_degree, "rows_resummed": rows_.value, "longest_row": longest_.value,
                "entries_resummed": entries_.value, "launches_per_iteration_for_the_resum": groups_.value,
                "plan_build_s": t_plan2, "pr_last_l1_change": pr2.global_diff(),
                "what": "the timed iteration with GDN_PR_SUM=reference GDN_PR_SUM_MIN_DEGREE=%d: behind the two kernels of the pull the "
                        "rows of that many in-edges are summed again in the reference's order -- fp32, one addition per in-edge, CSR "
                        "order (src/pr/omp_base.cc:27-30): their contributions staged in row order through LDS slices of the contribution "
                        "vector, then summed by scans of parity functions (csrc/gdn_seqsum.hpp), the longest rows on a workgroup each; "
                        "scores / next contributions / L1 change of those rows are rewritten.  frac = SURVEY 8d's bytes of the plain "
                        "iteration over this time." % args.refsum_min_degree}
            log(f"[bench] pr_reference_sum: {out['pr_reference_sum']}")
            be2.close()
            del pr2, be2
        except Exception as e:
            log(f"[bench] pr_reference_sum skipped: {e}")
        finally:
            L.gdn_option_set(b"GDN_PR_SUM", None)
            L.gdn_option_set(b"GDN_PR_SUM_MIN_DEGREE", None)

    # ---- BFS GTEPS on the same graph (single GPU: the N = 1 run carries it; BFS stays single-GPU per north_star, and the
    # other ranks of an N > 1 job would only wait in the final barrier for it), outside the timed region
    if rank == 0 and world == 1 and not args.no_bfs:
        try:
            # sources: the first vertices with out-degree > 0; every source is searched several times
            nz = torch.nonzero(out_degree[:1 << 16] > 0)[:4].flatten().tolist()
            dist_buf = torch.empty(m, dtype=torch.int32, device=device)
            t0 = time.time()
            bplan = C.c_void_p()
            _cabi.check(L.gdn_bfs_plan_create(g_out, g_in, 1, C.byref(bplan)))
            t_bplan = time.time() - t0
            runs = []
            per_src = max(1, -(-args.reps // max(len(nz[:3]), 1)))
            for s in nz[:3]:
                for _ in range(per_src):
                    st = _cabi.GdnStats()
                    _cabi.check(L.gdn_bfs_run(bplan, int(s), C.c_void_p(dist_buf.data_ptr()), C.byref(st)))
                    if st.edges_traversed > nnz // 100 and st.solve_ms > 0:
                        reached = int((dist_buf != 1000000000).sum().item())
                        # SURVEY 8d BFS bytes: SUM_reached (16 + 8 outdeg) + 4 m
                        b = 16 * reached + 8 * st.edges_traversed + 4 * m
                        runs.append({"source": int(s), "ms": st.solve_ms, "levels": st.iterations,
                                     "edges_traversed": st.edges_traversed, "reached": reached,
                                     "gteps": st.edges_traversed / (st.solve_ms * 1e-3) / 1e9,
                                     "gbs": b / (st.solve_ms * 1e-3) / 1e9, "bytes": b})
                log(f"[bench] BFS from {s}: {runs[-1] if runs else 'too small'}")
            # what of that time is the per-search initialisation (the 4 m-byte fill of the distances, the bitmap clear, the seed):
            # every BFSSolver of the reference does it in FRONT of its Timer (src/bfs/main.cc:21, linear_base.cu:50-63,
            # omp_beamer.cc:119-134); here it is inside solve_ms.  Two events around it in a few extra searches (GDN_BFS_TIME_INIT)
            init_ms, finish_ms = [], []
            try:
                _cabi.check(L.gdn_option_set(b"GDN_BFS_TIME_INIT", b"1"))
                for s in nz[:3]:
                    for _ in range(2):
                        st = _cabi.GdnStats()
                        _cabi.check(L.gdn_bfs_run(bplan, int(s), C.c_void_p(dist_buf.data_ptr()), C.byref(st)))
                        init_ms.append(st.prep_ms)
                        finish_ms.append(st.last_error)
            finally:
                L.gdn_option_set(b"GDN_BFS_TIME_INIT", None)
            L.gdn_bfs_plan_free(bplan)
            if runs:
                # the line leads with the MEDIAN run (the search whose GTEPS is the median of all runs over the three
                # sources); the best one is kept beside it
                best = max(runs, key=lambda r: r["gteps"])
                by = sorted(runs, key=lambda r: r["gteps"])
                med = by[(len(by) - 1) // 2]
                init_med = sorted(init_ms)[len(init_ms) // 2] if init_ms else None
                finish_med = sorted(finish_ms)[len(finish_ms) // 2] if finish_ms else None
                # what the reference's Timer does not see: the initialisation in front of the search (init_med) and, of the
                # closing pass, only the "unreached" fill it absorbed -- modelled as 4 m bytes at the 6.3 TB/s a plain stream
                # reaches (MI355X_MICROARCH.md), never more than the pass took.  The depths of the kept heavy levels, which the
                # pass also writes, stay inside: the reference writes them inside its Timer
                fill_model = min(4.0 * m / 6.3e12 * 1e3, finish_med) if finish_med else 0.0
                off_timer = (init_med + fill_model) if init_med is not None else None
                out["bfs"] = dict(med, plan_build_s=t_bplan, ms_stats=med_min([r["ms"] for r in runs]), runs=len(runs),
                                  init_ms_inside_solve=init_med, depth_finish_pass_ms=finish_med, unreached_fill_modelled_ms=fill_model,
                                  gteps_on_the_reference_timer=(med["edges_traversed"] / ((med["ms"] - off_timer) * 1e-3) / 1e9
                                                                if off_timer is not None and med["ms"] > off_timer else None),
                                  timer_note="ms / gteps include the per-search initialisation (bitmaps cleared, seed) and the closing pass "
                                             "that writes the distances of the kept heavy levels AND the 'unreached' value (it replaces the "
                                             "4 m-byte fill the reference's BFSSolver does in front of its Timer); "
                                             "gteps_on_the_reference_timer leaves out init_ms_inside_solve and unreached_fill_modelled_ms "
                                             "(4 m bytes at 6.3 TB/s, capped by the pass) only -- the rest of depth_finish_pass_ms writes "
                                             "real depths and stays inside, as in the reference",
                                  gteps_median=med["gteps"], gteps_best=best["gteps"],
                                  roofline={"bound": "hbm", "achieved": med["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "speed_vs_model": med["gbs"] / HBM_PEAK_GBS, "speed_vs_model_best_run": best["gbs"] / HBM_PEAK_GBS,
                                            "algorithmic_bytes": med["bytes"],
                                            "model": "SUM_reached (16 + 8 outdeg) + 4 m (SURVEY 8d)",
                                            "note": "the model charges every out-edge of the reached part (a top-down "
                                                    "search); the direction-optimizing search skips most of them in "
                                                    "its bottom-up levels, so frac can exceed 1 -- it compares with a "
                                                    "search that walks all edges, it is not an HBM utilisation: that is `frac` "
                                                    "(= frac_traffic, counter bytes / time / peak; null where no counter session "
                                                    "of this scale is committed) (per-level figures: profiles/r03_bfs_bottom_up.txt)"})
                # counter traffic was taken on the FIRST source (tools/traffic_run.py): its own time is the divisor
                first = sorted(r["ms"] for r in runs if r["source"] == runs[0]["source"])
                attach_traffic(out["bfs"]["roofline"], "bfs", args.scale, first[(len(first) - 1) // 2] * 1e-3)
                out["bfs"]["roofline"]["traffic_of_source"] = runs[0]["source"]
                out["bfs"]["roofline"]["frac"] = out["bfs"]["roofline"]["frac_traffic"]  # the utilisation
                out["bfs"]["ms_by_source"] = {str(src_): sorted(r["ms"] for r in runs if r["source"] == src_) for src_ in sorted({r["source"] for r in runs})}
                out["gteps_bfs"] = med["gteps"]
                out["gteps_bfs_best"] = best["gteps"]
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
            scores[:] = np.float32(1.0) / np.float32(m)  # (the probe overwrote its rows: the sample starts from 1/m again)
            tc = time.time()
            orc.pr_iterate(gi, h_deg, scores, 1, row_lo=0, row_hi=row_hi)
            tc = time.time() - tc
            e_sample = int(h_rp[row_hi])
            out["cpu_baseline"] = {"value": e_sample / tc, "unit": "edges/s", "cores": cores, "kind": "port",
                                   "physical_cores": n_phys, "logical_cpus": os.cpu_count(),
                                   "omp_proc_bind": os.environ.get("OMP_PROC_BIND"), "omp_places": os.environ.get("OMP_PLACES"),
                                   "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS"),
                                   "sample": "1 pull iteration over rows [0,%d) of the same RMAT-%d graph "
                                             "(%d edges, %.1f%% of the graph) incl. the contrib pass over all "
                                             "vertices, OpenMP restatement of src/pr/omp_base.cc:23-34"
                                             % (row_hi, args.scale, e_sample, 100.0 * e_sample / nnz),
                                   "seconds": tc}
            log(f"[bench] cpu baseline: {out['cpu_baseline']} (download+prep {time.time() - t1 - tc:.1f} s)")
            # parity of the timed plan with that CPU iteration on the rows it covered (the same start vector 1/m): how many
            # rows lie beyond north_star's 1e-4 -- rows with >= 10^4 in-edges, where the reference's sequential fp32 sum
            # drifts and the plan's exact fixed-point sum does not (bounded by tests/test_gpu_configs.py)
            try:
                if be.layout == 1 and not multi:
                    pp = lambda t: C.c_void_p(t.data_ptr())
                    ms_ = C.c_int32()
                    _cabi.check(L.gdn_pr_plan_state_size(be.plan, C.byref(ms_)))
                    start = torch.full((m,), 1.0 / m, dtype=torch.float32, device=device)
                    state = torch.empty(ms_.value, dtype=torch.float32, device=device)
                    cc = [torch.zeros(ms_.value + 4, dtype=torch.float32, device=device) for _ in range(2)]
                    dd = torch.zeros(1, dtype=torch.float64, device=device)
                    got = torch.empty(m, dtype=torch.float32, device=device)
                    _cabi.check(L.gdn_pr_import_dev(be.plan, pp(start), pp(state), 0.85, None))
                    _cabi.check(L.gdn_pr_contrib_dev(be.plan, pp(state), pp(cc[0]), None))
                    _cabi.check(L.gdn_pr_pull_dev(be.plan, pp(cc[0]), pp(state), pp(cc[1]), pp(dd), 0.85, None))
                    _cabi.check(L.gdn_pr_export_dev(be.plan, pp(state), pp(got), 0.85, None))
                    torch.cuda.synchronize()
                    g_rows = got[:row_hi].cpu().numpy()
                    rel = np.abs(g_rows - scores[:row_hi]) / scores[:row_hi]
                    off = np.nonzero(rel >= 1e-4)[0]
                    indeg = np.diff(h_rp[:row_hi + 1].astype(np.int64))
                    out["parity_note"] = {
                        "rows_compared": int(row_hi), "tolerance": 1e-4, "rows_beyond_tolerance": int(len(off)),
                        "max_rel": float(rel.max()) if row_hi else 0.0,
                        "min_in_degree_of_those_rows": int(indeg[off].min()) if len(off) else None,
                        "what": "one pull iteration of the timed plan vs the CPU restatement of src/pr/omp_base.cc:23-34 "
                                "from scores 1/m, on the rows of the cpu_baseline sample; rows beyond 1e-4 are hub rows "
                                "(>= 10^4 in-edges): the reference adds their contributions one by one in fp32, the plan "
                                "adds them exactly (DESIGN 5); tests/test_gpu_configs.py asserts <= 500 such rows of all "
                                "2^27 and max_rel <= 2.5e-3, and that the GPU value is the one an fp64 evaluation gives"}
                    log(f"[bench] parity note: {out['parity_note']}")
                    # ---- the same at CONVERGENCE: the timed plan iterated to epsilon 1e-4 against the
                    # oracle's own solve of the whole graph (src/pr/omp_base.cc:8-42, all rows, OpenMP on the host cores)
                    if not args.no_converged_parity:
                        tcv = time.time()
                        want, it_cpu, trace_cpu = orc.pr(gi, h_deg)
                        t_cpu_solve = time.time() - tcv
                        _cabi.check(L.gdn_pr_import_dev(be.plan, pp(start), pp(state), 0.85, None))
                        dead = C.c_double(0)
                        _cabi.check(L.gdn_pr_import_diff(be.plan, C.byref(dead)))
                        _cabi.check(L.gdn_pr_contrib_dev(be.plan, pp(state), pp(cc[0]), None))
                        trace_gpu, it_gpu = [], 0
                        for k in range(100):
                            _cabi.check(L.gdn_pr_pull_dev(be.plan, pp(cc[k & 1]), pp(state), pp(cc[(k + 1) & 1]), pp(dd), 0.85, None))
                            trace_gpu.append(float(dd.item()) + (dead.value if k == 0 else 0.0))
                            it_gpu = k + 1
                            if trace_gpu[-1] < 1e-4:
                                break
                        _cabi.check(L.gdn_pr_export_dev(be.plan, pp(state), pp(got), 0.85, None))
                        torch.cuda.synchronize()
                        g_all = got.cpu().numpy()
                        relc = np.abs(g_all - want) / want
                        offc = np.nonzero(relc >= 1e-4)[0]
                        indeg_all = np.diff(h_rp.astype(np.int64))
                        nt = min(len(trace_gpu), len(trace_cpu))
                        out["parity_note"]["converged"] = {
                            "iterations_gpu": it_gpu, "iterations_cpu": int(it_cpu), "rows_compared": int(m),
                            "rows_beyond_tolerance_converged": int(len(offc)), "max_rel_converged": float(relc.max()),
                            "min_in_degree_of_those_rows": int(indeg_all[offc].min()) if len(offc) else None,
                            "trace_max_rel_diff": float(max(abs(a - b) / b for a, b in zip(trace_gpu[:nt], trace_cpu[:nt]))) if nt else None,
                            "cpu_solve_s": t_cpu_solve,
                            "what": "the timed plan iterated from 1/m until the L1 change < 1e-4 vs the CPU restatement of "
                                    "src/pr/omp_base.cc:8-42 solving the whole graph; tests/test_gpu_fullsize.py asserts the "
                                    "same through the PRSolver drop-in, and that GDN_PR_SUM=reference on the rows of >= 10^4 "
                                    "in-edges leaves no row beyond 1e-4"}
                        log(f"[bench] converged parity: {out['parity_note']['converged']}")
                        del want, g_all, relc
                    del start, state, cc, got
            except Exception as e:
                log(f"[bench] parity note skipped: {e}")
            del h_rp, h_ci, gi, scores
        except Exception as e:
            log(f"[bench] cpu baseline skipped: {e}")
            out["cpu_baseline"] = None

    # ---- CPU baseline of an N > 1 job (north_star: "the host-OpenMP baseline ... in the same run"): rank 0, its share of the
    # host cores (OMP_NUM_THREADS above), the oracle's pull iteration on a bounded row sample of ITS shard -- the shard's rows
    # in the padded vertex space, the contrib pass over all of that space, like the N = 1 leg
    if want_cpu_multi and deg_state_host is not None:
        try:
            from oracle import binding as orc
            t1 = time.time()
            sm_, snnz_ = sm.value, snnz.value
            h_rp = np.empty(sm_ + 1, np.uint64)
            h_ci = np.empty(max(snnz_, 1), np.int32)
            _cabi.check(L.gdn_graph_download(shard, h_rp.ctypes.data_as(C.c_void_p), h_ci.ctypes.data_as(C.c_void_p)))
            rp_full = np.empty(m_space + 1, np.uint64)
            rp_full[:lo] = 0
            rp_full[lo:lo + sm_ + 1] = h_rp
            rp_full[lo + sm_ + 1:] = h_rp[-1]
            deg_pad = np.ones(m_space, np.int32)  # (pad slots: no edge names them)
            for r in range(world):
                deg_pad[r * chunk:r * chunk + (bl[r + 1] - bl[r])] = deg_state_host[bl[r]:bl[r + 1]]
            gi = graphio.CSR(m_space, rp_full, h_ci[:snnz_])
            scores = np.full(m_space, np.float32(1.0) / np.float32(m), np.float32)
            # (torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks unless the caller set it: rank 0 takes its share of
            # the host's physical cores for this leg; GDN_BENCH_CPU_THREADS overrides)
            orc.set_num_threads(int(os.environ.get("GDN_BENCH_CPU_THREADS", max(1, n_phys // max(n_ranks_here, 1)))))
            cores = orc.num_threads()
            probe_hi = lo + max(1, sm_ // 64)
            tp = time.time()
            orc.pr_iterate(gi, deg_pad, scores, 1, row_lo=lo, row_hi=probe_hi)
            tp = time.time() - tp
            frac = min(1.0, max(1.0 / 64, (args.cpu_seconds / max(tp, 1e-3)) / 64))
            row_hi = lo + max(1, int(sm_ * frac))
            scores[:] = np.float32(1.0) / np.float32(m)
            tc = time.time()
            orc.pr_iterate(gi, deg_pad, scores, 1, row_lo=lo, row_hi=row_hi)
            tc = time.time() - tc
            e_sample = int(h_rp[row_hi - lo])
            out["cpu_baseline"] = {"value": e_sample / tc, "unit": "edges/s", "cores": cores, "kind": "port",
                                   "physical_cores": n_phys, "logical_cpus": os.cpu_count(), "ranks_on_this_host": n_ranks_here,
                                   "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS"),
                                   "sample": "rank 0 of %d, its share of the host's cores: 1 pull iteration over the first %d rows of ITS "
                                             "shard of the same RMAT-%d graph (%d edges, %.1f%% of the shard, %.2f%% of the graph) incl. the "
                                             "contrib pass over the whole padded vertex space, OpenMP restatement of "
                                             "src/pr/omp_base.cc:23-34" % (world, row_hi - lo, args.scale, e_sample,
                                                                            100.0 * e_sample / max(snnz_, 1), 100.0 * e_sample / max(nnz, 1)),
                                   "seconds": tc}
            log(f"[bench] cpu baseline (rank 0 of {world}): {out['cpu_baseline']} (download+prep {time.time() - t1 - tc:.1f} s)")
            del h_rp, h_ci, rp_full, deg_pad, gi, scores
        except Exception as e:
            log(f"[bench] cpu baseline skipped: {e}")
            out["cpu_baseline"] = None

    # ---- BASELINE configs 3 and 4 on the same box (rank 0, N=1, outside the timed region): SpMV on RMAT-25 and TC on
    # symmetrized RMAT-23 (the Orkut-sized stand-in: com-Orkut is not in the repository, datasets/test.mk:8 is a wget line)
    if rank == 0 and world == 1 and not args.no_extras:
        import bench_extras as bx  # (the blocks beside the headline: bench_extras.py)
        be.close()
        del pr, be
        L.gdn_graph_free(g_out)
        L.gdn_graph_free(g_in)
        g_out = g_in = None
        torch.cuda.empty_cache()
        try:
            out["spmv"] = bx.bench_spmv(L, _cabi, graphio, torch, np, device, args)
        except Exception as e:
            log(f"[bench] spmv block skipped: {e}")
        try:
            out["tc"] = bx.bench_tc(L, _cabi, graphio, torch, np, device, args)
        except Exception as e:
            log(f"[bench] tc block skipped: {e}")
        try:
            out["traversal"] = bx.bench_traversal(L, _cabi, graphio, torch, np, device, args)
        except Exception as e:
            log(f"[bench] traversal block skipped: {e}")
        try:
            out["pr_oneshot"] = bx.bench_pr_oneshot(L, _cabi, graphio, np, args)
        except Exception as e:
            log(f"[bench] pr_oneshot block skipped: {e}")
        try:
            out["standins"] = bx.bench_standins(L, _cabi, graphio, torch, np, device, args)
        except Exception as e:
            log(f"[bench] stand-in block skipped: {e}")

    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
