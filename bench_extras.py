#!/usr/bin/env python3
"""bench_extras.py -- the blocks that ride on bench.py's N = 1 line OUTSIDE the PageRank timed region (split off bench.py in round
6): the PRSolver drop-in on an LJ-sized graph (pr_oneshot), BASELINE config 3 (fp32 SpMV on RMAT-25: spmv), config 4's stand-in
(triangle count on symmetrized RMAT-23: tc), the stand-ins shaped like configs 2 and 4 (standins) and SSSP / CC on RMAT-24
(traversal) -- each with median + min over >= 10 repetitions and its own roofline block.  bench.py imports this module only
when the blocks are wanted (N = 1, no --no-extras); nothing here touches the headline number."""
import ctypes as C
import json
import os
import time

from bench import HBM_PEAK_GBS, ROOT, attach_traffic, log, med_min  # noqa: F401  (the helpers of the headline script)


def bench_pr_oneshot(L, _cabi, graphio, np, args):
    """The PRSolver drop-in (gdn_pr: host arrays in, one call) on an LJ-sized graph (RMAT-22, BASELINE config 2's
    stand-in): time to convergence WITH the layout the call builds, for the merge-path and the blocked layout and for
    what the call picks on its own (by predicted wall time, gdn_pr.hip)."""
    go, gi = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build(22, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
    m, nnz = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(gi, C.byref(m), C.byref(nnz), None, None))
    m, nnz = m.value, nnz.value
    rp, ci = np.empty(m + 1, np.uint64), np.empty(nnz, np.int32)
    _cabi.check(L.gdn_graph_download(gi, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p)))
    orp = np.empty(m + 1, np.uint64)
    oci = np.empty(nnz, np.int32)
    _cabi.check(L.gdn_graph_download(go, orp.ctypes.data_as(C.c_void_p), oci.ctypes.data_as(C.c_void_p)))
    deg = np.diff(orp.astype(np.int64)).astype(np.int32)
    del oci, orp
    L.gdn_graph_free(go)
    L.gdn_graph_free(gi)
    rec = {"workload": "gdn_pr (PRSolver drop-in, one call on host arrays) to epsilon 1e-4, R-MAT scale 22 avg degree 16",
           "vertices": m, "edges": nnz}
    for name, lay in (("csr", b"csr"), ("pb", b"pb"), ("auto", None)):
        _cabi.check(L.gdn_option_set(b"GDN_PR_LAYOUT", lay))
        best = None
        for _ in range(3):
            scores = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
            st = _cabi.GdnStats()
            _cabi.check(L.gdn_pr(m, nnz, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), deg.ctypes.data_as(C.c_void_p),
                                 scores.ctypes.data_as(C.c_void_p), C.c_float(0.85), C.c_double(1e-4), 100, C.byref(st)))
            cur = {"iterations": st.iterations, "solve_ms": st.solve_ms, "prep_ms": st.prep_ms, "h2d_ms": st.h2d_ms,
                   "solve_plus_prep_ms": st.solve_ms + st.prep_ms}
            if best is None or cur["solve_plus_prep_ms"] < best["solve_plus_prep_ms"]:
                best = cur
        rec[name] = best
    L.gdn_option_set(b"GDN_PR_LAYOUT", None)
    rec["auto_picked"] = "csr" if abs(rec["auto"]["prep_ms"] - rec["csr"]["prep_ms"]) < abs(rec["auto"]["prep_ms"] - rec["pb"]["prep_ms"]) else "pb"
    log(f"[bench] pr_oneshot: {rec}")
    return rec


def bench_spmv(L, _cabi, graphio, torch, np, device, args):
    """fp32 SpMV y += A x on RMAT-<spmv-scale> x16, Ax and x ~ U(0,1) (src/spmv/main.cc:27-40 fills constants,
    which makes the gather value-degenerate): the resident plan (layout AUTO = propagation blocking at this size) and the
    one-shot drop-in gdn_spmv on host arrays.  Reported like src/spmv/omp_base.cc:37-40: ms, GFLOP/s, GB/s."""
    g_out, g_in = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build(args.spmv_scale, 16, graphio.K_RAND_SEED, 1, C.byref(g_out), C.byref(g_in)))
    L.gdn_graph_free(g_out)
    m, nnz = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(g_in, C.byref(m), C.byref(nnz), None, None))
    m, nnz = m.value, nnz.value
    gen = torch.Generator(device=device)
    gen.manual_seed(25)
    Ax = torch.rand(nnz, dtype=torch.float32, device=device, generator=gen)
    x = torch.rand(m, dtype=torch.float32, device=device, generator=gen)
    y = torch.zeros(m, dtype=torch.float32, device=device)
    t0 = time.time()
    plan = C.c_void_p()
    _cabi.check(L.gdn_spmv_plan_create(g_in, C.c_void_p(Ax.data_ptr()), _cabi.GDN_LAYOUT_AUTO, C.byref(plan)))
    torch.cuda.synchronize()
    t_plan = time.time() - t0
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def mul():
        _cabi.check(L.gdn_spmv_dev(plan, C.c_void_p(Ax.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), s))
    for _ in range(2):
        mul()
    torch.cuda.synchronize()
    reps = max(args.reps, 10)
    _cabi.check(L.gdn_spmv_plan_kernel_time(plan, 1, reps, None, None))
    wall = []
    for _ in range(reps):
        t1 = time.perf_counter()
        mul()
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t1) * 1e3)
    tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
    _cabi.check(L.gdn_spmv_plan_kernel_time(plan, 0, 0, tot, C.byref(n)))
    _cabi.check(L.gdn_spmv_plan_check(plan))
    k_ms = (tot[0] + tot[1]) / max(n.value, 1)
    nbytes = int(L.gdn_spmv_bytes(plan))
    nh, nt, te = C.c_int32(0), C.c_int32(0), C.c_uint64(0)
    _cabi.check(L.gdn_spmv_plan_tiers(plan, C.byref(nh), C.byref(nt), C.byref(te)))
    L.gdn_spmv_plan_free(plan)
    gbs = nbytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    rec = {"workload": "SpMV fp32 y += A x, R-MAT scale %d avg degree 16, Ax and x ~ U(0,1)" % args.spmv_scale,
           "rows": m, "nnz": nnz, "plan_build_s": t_plan, "ms": med_min(wall), "kernel_ms": k_ms,
           "kernel_ms_parts": [tot[0] / max(n.value, 1), tot[1] / max(n.value, 1)],
           "gflops": 2.0 * nnz / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0,
           "record_tier_nonzeros": int(te.value),
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": nbytes, "model": "8(m+1) + 12 nnz + 8 m (SURVEY 8d)",
                        "kernel": "pb_expand_scaled_kernel + pb_accumulate_kernel<SpmvOp>"}}
    attach_traffic(rec["roofline"], "spmv", args.spmv_scale, k_ms * 1e-3)
    # the one-shot drop-in on host arrays (what SpmvSolver binds to): upload, whatever it builds, one multiply
    try:
        h_rp, h_ci = np.empty(m + 1, np.uint64), np.empty(nnz, np.int32)
        _cabi.check(L.gdn_graph_download(g_in, h_rp.ctypes.data_as(C.c_void_p), h_ci.ctypes.data_as(C.c_void_p)))
        h_Ax, h_x, h_y = Ax.cpu().numpy(), x.cpu().numpy(), np.zeros(m, np.float32)
        shots = []
        for _ in range(2):  # the second call on the same arrays may hit what the first one built
            st = _cabi.GdnStats()
            t1 = time.perf_counter()
            _cabi.check(L.gdn_spmv(m, nnz, h_rp.ctypes.data_as(C.c_void_p), h_ci.ctypes.data_as(C.c_void_p),
                                   h_Ax.ctypes.data_as(C.c_void_p), h_x.ctypes.data_as(C.c_void_p),
                                   h_y.ctypes.data_as(C.c_void_p), C.byref(st)))
            shots.append({"wall_ms": (time.perf_counter() - t1) * 1e3, "solve_ms": st.solve_ms, "prep_ms": st.prep_ms,
                          "h2d_ms": st.h2d_ms,
                          "frac_solve": nbytes / (st.solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if st.solve_ms > 0 else 0.0,
                          "frac_solve_plus_prep": nbytes / ((st.solve_ms + st.prep_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS
                          if st.solve_ms + st.prep_ms > 0 else 0.0})
        rec["oneshot_gdn_spmv"] = shots
        # the same call with GDN_SPMV_ONESHOT=solve: the blocked layout, its build in prep_ms -- the timing boundary of the
        # reference's own blocked solver (segmenting() in front of the Timer, src/spmv/partition.cu:206,269-291); wall time
        # (prep + solve) is worse than the default's, which is why it is an option
        _cabi.check(L.gdn_option_set(b"GDN_SPMV_ONESHOT", b"solve"))
        try:
            st = _cabi.GdnStats()
            h_y2 = np.zeros(m, np.float32)
            _cabi.check(L.gdn_spmv(m, nnz, h_rp.ctypes.data_as(C.c_void_p), h_ci.ctypes.data_as(C.c_void_p),
                                   h_Ax.ctypes.data_as(C.c_void_p), h_x.ctypes.data_as(C.c_void_p),
                                   h_y2.ctypes.data_as(C.c_void_p), C.byref(st)))
            rec["oneshot_gdn_spmv_blocked_layout"] = {
                "option": "GDN_SPMV_ONESHOT=solve", "solve_ms": st.solve_ms, "prep_ms": st.prep_ms, "h2d_ms": st.h2d_ms,
                "frac_solve": nbytes / (st.solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if st.solve_ms > 0 else 0.0,
                "frac_solve_plus_prep": nbytes / ((st.solve_ms + st.prep_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "max_rel_diff_vs_default_call": float((np.abs(h_y2 - h_y / 2) / np.maximum(np.abs(h_y2), 1e-30)).max())}
        finally:
            L.gdn_option_set(b"GDN_SPMV_ONESHOT", None)
    except Exception as e:
        log(f"[bench] spmv one-shot skipped: {e}")
    L.gdn_graph_free(g_in)
    log(f"[bench] spmv: {rec}")
    return rec


def bench_tc(L, _cabi, graphio, torch, np, device, args):
    """Triangle count on symmetrized RMAT-<tc-scale> x16 (RMAT-23: 129 M DAG edges, the size of com-Orkut's 117 M).
    TEPS = DAG edges / time as src/tc/gpu_base.cu:60; bytes = SURVEY 8d's merge-equivalent model."""
    g_out, sym, dag = C.c_void_p(), C.c_void_p(), C.c_void_p()
    what = "symmetrized R-MAT scale %d avg degree 16 (com-Orkut-sized stand-in)" % args.tc_scale
    orkut = os.path.join(ROOT, "datasets", "com-Orkut")
    if os.path.exists(orkut + ".mtx") or os.path.exists(orkut + ".meta.txt"):  # config 4's own input, when present
        g = graphio.read_mtx(orkut + ".mtx", True) if os.path.exists(orkut + ".mtx") else graphio.symmetrize(graphio.read_bin(orkut))
        _cabi.check(L.gdn_graph_upload(g.m, g.nnz, g.rowptr.ctypes.data_as(C.c_void_p), g.colidx.ctypes.data_as(C.c_void_p), C.byref(sym)))
        what = "com-Orkut (datasets/com-Orkut.*), symmetrized"
        del g
    else:
        _cabi.check(L.gdn_rmat_build(args.tc_scale, 16, graphio.K_RAND_SEED, 1, C.byref(g_out), None))
        _cabi.check(L.gdn_graph_symmetrize(g_out, C.byref(sym)))
        L.gdn_graph_free(g_out)
    t0 = time.time()
    _cabi.check(L.gdn_graph_orient(sym, C.byref(dag)))
    t_orient = time.time() - t0
    m, nnz, snnz = C.c_int32(), C.c_uint64(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(sym, None, C.byref(snnz), None, None))
    _cabi.check(L.gdn_graph_info(dag, C.byref(m), C.byref(nnz), None, None))
    L.gdn_graph_free(sym)
    nbytes = C.c_uint64(0)
    _cabi.check(L.gdn_tc_model_bytes(dag, C.byref(nbytes)))
    total, ms = C.c_uint64(0), []
    t0 = time.time()
    tplan = C.c_void_p()
    _cabi.check(L.gdn_tc_plan_create(dag, 1, C.byref(tplan)))  # what the reference does while loading (src/tc/main.cc:12)
    t_tplan = time.time() - t0
    for i in range(max(args.reps, 10) + 1):
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_tc_plan_count(tplan, C.byref(total), C.byref(st)))
        if i:
            ms.append(st.solve_ms)
    walked_plan = C.c_uint64(0)  # forward count: the list elements walked around middle vertices below the core
    _cabi.check(L.gdn_tc_plan_walked_elements(tplan, C.byref(walked_plan)))
    L.gdn_tc_plan_free(tplan)
    form = {0: "u-centric", 1: "v-centric", 2: "binary search", 3: "forward (rank-ordered DAG, walks start behind v)"}.get(st.reserved & 0xFF, "?")
    core_ranks = st.reserved >> 8  # forward count: the top ranks counted on the core bit matrix (tc_core_count_kernel)
    mm = med_min(ms)
    gbs = nbytes.value / (mm["median"] * 1e-3) / 1e9
    # what the kernel itself reads: ONE list per DAG edge (4 B x the probes of the formulation that ran) + the row's own
    # list + the offsets -- the merge-equivalent model counts both lists of every edge, so its fraction overstates the
    # memory rate; this one is the kernel's own list traffic, most of it served beyond L2
    probes = C.c_uint64 * 2
    pr_ = probes(0, 0)
    list_gbs = None
    if hasattr(L, "gdn_tc_probe_counts"):
        _cabi.check(L.gdn_tc_probe_counts(dag, pr_))
        # forward form: SUM_u C(d+(u), 2) = (SUM_u d+(u)^2 - nnz) / 2 list elements (the out-degrees do not depend on the labelling)
        walked = (pr_[1] - nnz.value) // 2 if (st.reserved & 0xFF) == 3 else (pr_[1] if st.reserved == 1 else pr_[0])
        read_b = 4 * walked + 12 * nnz.value + 16 * (m.value + 1)
        # with the core only the walks around middle vertices BELOW the top ranks are walks of lists (gdn_tc_plan_walked_elements);
        # the core kernel's row reads of the bit matrix are not in this figure (they are in `frac`, the counter bytes)
        if core_ranks:
            read_b = 4 * walked_plan.value + 12 * nnz.value + 16 * (m.value + 1)
        list_gbs = read_b / (mm["median"] * 1e-3) / 1e9
    rec = {"workload": "triangle count, %s, DAG orientation by degree (src/common/graph.cc:67)" % what,
           "vertices": m.value, "undirected_csr_entries": snnz.value, "dag_edges": nnz.value, "triangles": total.value,
           "orient_s": t_orient, "plan_build_s": t_tplan, "ms": mm, "gteps": nnz.value / (mm["median"] * 1e-3) / 1e9, "formulation": form,
           "core_ranks": core_ranks,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "speed_vs_model": gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": nbytes.value,
                        "model": "4 SUM_(u,v) (d+(u) + d+(v)) + 8 nnz_dag + 8(m+1) (SURVEY 8d, merge-equivalent)",
                        "kernel": "tc_count_kernel" + (" beside tc_core_count_kernel (the look-ups around the top %d ranks on a bit matrix, DESIGN 4.7)" % core_ranks if core_ranks else ""),
                        "note": "the model counts what a MERGE intersect reads (both lists of every DAG edge, whole); the forward "
                                "count walks one list per edge from behind v on -- a quarter of those elements -- so "
                                "speed_vs_model is a speed relative to the merge formulation and can exceed 1; the HBM utilisation "
                                "is `frac` (= frac_traffic: counter bytes / time / peak, null where no counter session of this "
                                "scale is committed)",
                        "kernel_list_read_gbs": list_gbs,
                        "kernel_list_read_frac": list_gbs / HBM_PEAK_GBS if list_gbs else None,
                        "list_elements_walked": walked_plan.value if (st.reserved & 0xFF) == 3 else None,
                        "kernel_list_read_model": "4 B x the list elements the formulation that ran walks (one list per DAG "
                                                  "edge, from behind v in the forward form) + 12 nnz_dag + 16(m+1): what the "
                                                  "kernel requests, not what the model credits; counters: profiles/r03_tc_pmc.md"}}
    attach_traffic(rec["roofline"], "tc", args.tc_scale, mm["median"] * 1e-3)
    rec["roofline"]["frac"] = rec["roofline"]["frac_traffic"]
    # the one-shot drop-in on the resident DAG (what TCSolver binds to: no plan handed in; the preparation -- rank order, transpose,
    # walk starts, core matrix -- is stats.prep_ms, the count stats.solve_ms)
    try:
        shots = []
        for _ in range(3):
            so, t1 = _cabi.GdnStats(), C.c_uint64(0)
            _cabi.check(L.gdn_tc_dev(dag, 1, C.byref(t1), C.byref(so)))
            shots.append({"prep_ms": so.prep_ms, "solve_ms": so.solve_ms, "same_count": t1.value == total.value})
        rec["oneshot_gdn_tc_dev"] = shots
    except Exception as e:
        log(f"[bench] tc one-shot skipped: {e}")
    # A/B: the forward count without its core (round 3's kernel: every look-up a list element against the hash set)
    if core_ranks:
        try:
            _cabi.check(L.gdn_option_set(b"GDN_TC_CORE", b"0"))
            tn, msn = C.c_uint64(0), []
            pn_ = C.c_void_p()
            _cabi.check(L.gdn_tc_plan_create(dag, 1, C.byref(pn_)))
            for i in range(5):
                sn = _cabi.GdnStats()
                _cabi.check(L.gdn_tc_plan_count(pn_, C.byref(tn), C.byref(sn)))
                if i:
                    msn.append(sn.solve_ms)
            L.gdn_tc_plan_free(pn_)
            rec["ab_forward_without_core"] = {"ms": med_min(msn), "same_count": tn.value == total.value}
        except Exception as e:
            log(f"[bench] tc core A/B skipped: {e}")
        finally:
            L.gdn_option_set(b"GDN_TC_CORE", None)
    # A/B: round 2's default (hash set, u- or v-centric on the reference's orientation, whichever probes less)
    try:
        _cabi.check(L.gdn_option_set(b"GDN_TC_FORM", b"a"))
        ta, msa = C.c_uint64(0), []
        pa_ = C.c_void_p()
        _cabi.check(L.gdn_tc_plan_create(dag, 1, C.byref(pa_)))
        for i in range(4):
            sa = _cabi.GdnStats()
            _cabi.check(L.gdn_tc_plan_count(pa_, C.byref(ta), C.byref(sa)))
            if i:
                msa.append(sa.solve_ms)
        L.gdn_tc_plan_free(pa_)
        rec["ab_hash_set_unpruned"] = {"ms": med_min(msa), "same_count": ta.value == total.value,
                                       "formulation": {0: "u-centric", 1: "v-centric"}.get(sa.reserved, "?")}
    except Exception as e:
        log(f"[bench] tc unpruned A/B skipped: {e}")
    finally:
        L.gdn_option_set(b"GDN_TC_FORM", None)
    # A/B: the north star's wave-per-edge binary-search intersect (GDN_TC_FORM=bs) on the same DAG, same count
    try:
        _cabi.check(L.gdn_option_set(b"GDN_TC_FORM", b"bs"))
        tb, msb = C.c_uint64(0), []
        pb_ = C.c_void_p()
        _cabi.check(L.gdn_tc_plan_create(dag, 1, C.byref(pb_)))
        for i in range(3):
            sb = _cabi.GdnStats()
            _cabi.check(L.gdn_tc_plan_count(pb_, C.byref(tb), C.byref(sb)))
            if i:
                msb.append(sb.solve_ms)
        L.gdn_tc_plan_free(pb_)
        rec["ab_binary_search_intersect"] = {"ms": med_min(msb), "same_count": tb.value == total.value,
                                             "gteps": nnz.value / (med_min(msb)["median"] * 1e-3) / 1e9,
                                             "kernel": "tc_bs_count_kernel (one wavefront per DAG edge, 64 LDS pivots; "
                                                       "src/tc/gpu_base.cu:11-23 re-cut for wave64)"}
    except Exception as e:
        log(f"[bench] tc binary-search A/B skipped: {e}")
    finally:
        L.gdn_option_set(b"GDN_TC_FORM", None)
    L.gdn_graph_free(dag)
    log(f"[bench] tc: {rec}")
    return rec


def bench_standins(L, _cabi, graphio, torch, np, device, args):
    """BASELINE configs 2 and 4 on graphs SHAPED like their inputs (graphio.LJ_LIKE / ORKUT_LIKE: gdn_rmat_build_ex with milder
    quadrant probabilities and the ids without an edge dropped; soc-LiveJournal1 and com-Orkut themselves are wget lines,
    datasets/test.mk:5,8): PageRank pull iteration on the LJ-like graph (resident plan, as the headline), triangle count on the
    symmetrized Orkut-like graph.  Every tuning decision of rounds 3-4 was made on Graph500 R-MAT; these say what it is worth
    on a flatter degree distribution without isolated vertices."""
    rec = {}
    shrink = lambda r: dict(r, scale=r["scale"] - args.standin_shrink, n_edges=r["n_edges"] >> args.standin_shrink)
    # ---- PageRank, LJ-like
    r = shrink(graphio.LJ_LIKE)
    go, gi = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build_ex(r["scale"], r["n_edges"], *r["abc"], graphio.K_RAND_SEED, r["flags"], C.byref(go), C.byref(gi)))
    m, nnz = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(gi, C.byref(m), C.byref(nnz), None, None))
    m, nnz = m.value, nnz.value
    deg = torch.empty(m, dtype=torch.int32, device=device)
    _cabi.check(L.gdn_graph_degrees_dev(go, C.c_void_p(deg.data_ptr()), None))
    indeg = torch.empty(m, dtype=torch.int32, device=device)
    _cabi.check(L.gdn_graph_degrees_dev(gi, C.c_void_p(indeg.data_ptr()), None))
    L.gdn_graph_free(go)
    pp = lambda t: C.c_void_p(t.data_ptr())
    t0 = time.time()
    plan = C.c_void_p()
    _cabi.check(L.gdn_pr_plan_create(gi, pp(deg), m, 0, _cabi.GDN_LAYOUT_PB_SQUISHED, C.byref(plan)))
    t_plan = time.time() - t0
    ms_ = C.c_int32()
    _cabi.check(L.gdn_pr_plan_state_size(plan, C.byref(ms_)))
    start = torch.full((m,), 1.0 / m, dtype=torch.float32, device=device)
    state = torch.empty(ms_.value, dtype=torch.float32, device=device)
    cc = [torch.zeros(ms_.value + 4, dtype=torch.float32, device=device) for _ in range(2)]
    dd = torch.zeros(1, dtype=torch.float64, device=device)
    _cabi.check(L.gdn_pr_import_dev(plan, pp(start), pp(state), 0.85, None))
    _cabi.check(L.gdn_pr_contrib_dev(plan, pp(state), pp(cc[0]), None))
    reps = max(args.reps, 10)
    for k in range(3):
        _cabi.check(L.gdn_pr_pull_dev(plan, pp(cc[k & 1]), pp(state), pp(cc[(k + 1) & 1]), pp(dd), 0.85, None))
    torch.cuda.synchronize()
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 1, reps, None, None))
    for k in range(3, 3 + reps):
        _cabi.check(L.gdn_pr_pull_dev(plan, pp(cc[k & 1]), pp(state), pp(cc[(k + 1) & 1]), pp(dd), 0.85, None))
    tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_kernel_time(plan, 0, 0, tot, C.byref(n)))
    _cabi.check(L.gdn_pr_plan_check(plan))
    k_ms = (tot[0] + tot[1]) / max(n.value, 1)
    nbytes = int(L.gdn_pr_iter_bytes(plan))
    nh, he = C.c_int32(0), C.c_uint64(0)
    _cabi.check(L.gdn_pr_plan_hubs(plan, C.byref(nh), C.byref(he)))
    mt, msrc, me = C.c_int32(0), C.c_int32(0), C.c_uint64(0)
    _cabi.check(L.gdn_pr_plan_mid(plan, C.byref(mt), C.byref(msrc), C.byref(me)))
    L.gdn_pr_plan_free(plan)
    L.gdn_graph_free(gi)
    gbs = nbytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    rec["pr_lj_like"] = {"workload": "PageRank pull iteration, LJ-like stand-in (R-MAT scale %d, %d draws, a/b/c %s, ids without an "
                                     "edge dropped)" % (r["scale"], r["n_edges"], r["abc"]),
                         "vertices": m, "edges": nnz, "max_in_degree": int(indeg.max().item()), "max_out_degree": int(deg.max().item()),
                         "plan_build_s": t_plan, "kernel_ms": k_ms, "kernel_ms_parts": [tot[0] / max(n.value, 1), tot[1] / max(n.value, 1)],
                         "edges_per_s": nnz / (k_ms * 1e-3) if k_ms > 0 else 0.0,
                         "record_tiers": {"hub_sources": nh.value, "hub_edges": he.value, "mid_tiers": mt.value,
                                          "mid_sources": msrc.value, "mid_edges": me.value,
                                          "share_of_edges": (he.value + me.value) / max(nnz, 1)},
                         "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                      "algorithmic_bytes_per_launch": nbytes, "model": "8(m+1) + 8 nnz + 16 m (SURVEY 8d)"}}
    del deg, indeg, start, state, cc
    # ---- triangle count, Orkut-like
    r = shrink(graphio.ORKUT_LIKE)
    go, sym, dag = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build_ex(r["scale"], r["n_edges"], *r["abc"], graphio.K_RAND_SEED, r["flags"], C.byref(go), None))
    _cabi.check(L.gdn_graph_symmetrize(go, C.byref(sym)))
    L.gdn_graph_free(go)
    sm, snnz = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(sym, C.byref(sm), C.byref(snnz), None, None))
    sdeg = torch.empty(sm.value, dtype=torch.int32, device=device)
    _cabi.check(L.gdn_graph_degrees_dev(sym, C.c_void_p(sdeg.data_ptr()), None))
    _cabi.check(L.gdn_graph_orient(sym, C.byref(dag)))
    L.gdn_graph_free(sym)
    dn = C.c_uint64()
    _cabi.check(L.gdn_graph_info(dag, None, C.byref(dn), None, None))
    t0 = time.time()
    tplan = C.c_void_p()
    _cabi.check(L.gdn_tc_plan_create(dag, 1, C.byref(tplan)))
    t_tplan = time.time() - t0
    total, ms = C.c_uint64(0), []
    for i in range(reps + 1):
        st = _cabi.GdnStats()
        _cabi.check(L.gdn_tc_plan_count(tplan, C.byref(total), C.byref(st)))
        if i:
            ms.append(st.solve_ms)
    L.gdn_tc_plan_free(tplan)
    so, t1 = _cabi.GdnStats(), C.c_uint64(0)
    _cabi.check(L.gdn_tc_dev(dag, 1, C.byref(t1), C.byref(so)))
    L.gdn_graph_free(dag)
    mm = med_min(ms)
    rec["tc_orkut_like"] = {"workload": "triangle count, Orkut-like stand-in (R-MAT scale %d, %d draws, a/b/c %s, ids without an edge "
                                        "dropped, symmetrized)" % (r["scale"], r["n_edges"], r["abc"]),
                            "vertices": sm.value, "undirected_edges": snnz.value // 2, "max_degree": int(sdeg.max().item()),
                            "dag_edges": dn.value, "triangles": total.value, "plan_build_s": t_tplan, "ms": mm,
                            "gteps": dn.value / (mm["median"] * 1e-3) / 1e9, "formulation": st.reserved & 0xFF,
                            "core_ranks": st.reserved >> 8,
                            "oneshot_gdn_tc_dev": {"prep_ms": so.prep_ms, "solve_ms": so.solve_ms, "same_count": t1.value == total.value}}
    log(f"[bench] stand-ins: {rec}")
    return rec


def bench_traversal(L, _cabi, graphio, torch, np, device, args):
    """SSSP (resident plan: dense sweeps + fused light phases) and CC on R-MAT-<trav-scale> x16, directed.  SSSP bytes =
    the BFS model + 4 B of weight per reached edge: SUM_reached (16 + 12 outdeg) + 4 m; CC reports edges/s."""
    go, gi = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_rmat_build(args.trav_scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
    m, nnz = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(go, C.byref(m), C.byref(nnz), None, None))
    m, nnz = m.value, nnz.value
    deg = torch.empty(m, dtype=torch.int32, device=device)
    _cabi.check(L.gdn_graph_degrees_dev(go, C.c_void_p(deg.data_ptr()), None))
    src = int(torch.nonzero(deg[:1 << 16] > 0)[0].item())
    dist = torch.empty(m, dtype=torch.int32, device=device)
    reps = max(args.reps, 10)
    rec = {"workload": "R-MAT scale %d avg degree 16, directed" % args.trav_scale, "vertices": m, "edges": nnz, "source": src}
    gen = torch.Generator(device=device)
    gen.manual_seed(5)
    for name, w, delta in (("sssp_unit", torch.ones(nnz, dtype=torch.int32, device=device), 1),
                           ("sssp_u1_255_delta16", torch.randint(1, 256, (nnz,), dtype=torch.int32, device=device, generator=gen), 16)):
        plan = C.c_void_p()
        t0 = time.time()
        _cabi.check(L.gdn_sssp_plan_create(go, C.c_void_p(w.data_ptr()), 1, C.byref(plan)))
        t_plan = time.time() - t0
        ms = []
        for i in range(reps + 1):
            st = _cabi.GdnStats()
            _cabi.check(L.gdn_sssp_run(plan, src, delta, C.c_void_p(dist.data_ptr()), C.byref(st)))
            if i:
                ms.append(st.solve_ms)
        L.gdn_sssp_plan_free(plan)
        ab_sweeps = None
        if name == "sssp_unit" and st.reserved == 1:
            # equal weights: the plan solves through the BFS plan (round 5, gdn_sssp.hip); the same solve on the blocked
            # Bellman-Ford sweeps (what rounds 1-4 reported under this name) beside it
            try:
                _cabi.check(L.gdn_option_set(b"GDN_SSSP_UNIT_BFS", b"0"))
                p2 = C.c_void_p()
                _cabi.check(L.gdn_sssp_plan_create(go, C.c_void_p(w.data_ptr()), 1, C.byref(p2)))
                ms2 = []
                d2 = torch.empty(m, dtype=torch.int32, device=device)
                for i in range(6):
                    s2 = _cabi.GdnStats()
                    _cabi.check(L.gdn_sssp_run(p2, src, delta, C.c_void_p(d2.data_ptr()), C.byref(s2)))
                    if i:
                        ms2.append(s2.solve_ms)
                L.gdn_sssp_plan_free(p2)
                ab_sweeps = {"ms": med_min(ms2), "phases": s2.iterations, "same_distances": bool((d2 == dist).all().item())}
                del d2
            finally:
                L.gdn_option_set(b"GDN_SSSP_UNIT_BFS", None)
        reached = int((dist != 2147483647).sum().item())
        relaxed = int(st.last_error)  # SSSP: edges relaxed over the solve (include/gardenia_hip.h, gdn_stats)
        b = 16 * reached + 12 * st.edges_traversed + 4 * m
        b_relaxed = 16 * reached + 12 * relaxed + 4 * m  # "... x re-relaxation count (report edges relaxed)"
        mm = med_min(ms)
        sec = mm["median"] * 1e-3
        rec[name] = {"ms": mm, "phases": st.iterations, "edges_traversed": st.edges_traversed, "edges_relaxed": relaxed,
                     "re_relaxation": relaxed / max(st.edges_traversed, 1), "plan_build_s": t_plan,
                     "gteps": st.edges_traversed / sec / 1e9,
                     "roofline": {"bound": "hbm", "achieved": b / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": b / sec / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": b,
                                  "model": "SUM_reached (16 + 12 outdeg) + 4 m (SURVEY 8d: BFS bytes + 4 B weight per edge)",
                                  "frac_on_relaxed_edges": b_relaxed / sec / 1e9 / HBM_PEAK_GBS,
                                  "bytes_on_relaxed_edges": b_relaxed,
                                  "model_relaxed": "16 reached + 12 edges_relaxed + 4 m: every relaxation the solver made "
                                                   "(list passes: the out-edges of their list; a dense sweep: every edge)"}}
        if st.reserved == 1:  # solved as a BFS: the model charges every reached edge, the search skips most -- a speed, not a utilisation
            roof = rec[name]["roofline"]
            roof["speed_vs_model"] = roof.pop("frac")
            roof["frac"] = None
        if ab_sweeps is not None and st.reserved == 1:
            rec[name]["route"] = "equal weights: direction-optimising BFS plan on the transpose, depths x weight (plan_build_s includes the transpose)"
            rec[name]["ab_dense_sweeps"] = ab_sweeps
        else:
            attach_traffic(rec[name]["roofline"], "sssp_unit" if name == "sssp_unit" else "sssp_u255", args.trav_scale, sec)
        # the one-shot drop-in on the resident graph (what SSSPSolver binds to: no plan handed in; from 2^24 edges on the
        # call builds the blocked layout itself and reports it as prep_ms)
        try:
            shots = []
            for _ in range(3):
                so = _cabi.GdnStats()
                _cabi.check(L.gdn_sssp_dev(go, C.c_void_p(w.data_ptr()), src, delta, C.c_void_p(dist.data_ptr()), C.byref(so)))
                shots.append({"solve_ms": so.solve_ms, "prep_ms": so.prep_ms, "phases": so.iterations})
            rec[name]["oneshot_gdn_sssp_dev"] = shots
        except Exception as e:
            log(f"[bench] sssp one-shot skipped: {e}")
        del w
    comp = torch.empty(m, dtype=torch.int32, device=device)
    for name, rev in (("cc_with_reverse_graph", gi), ("cc_out_edges_only", None)):
        ms = []
        for i in range(reps + 1):
            st = _cabi.GdnStats()
            _cabi.check(L.gdn_cc_dev(go, rev, C.c_void_p(comp.data_ptr()), C.byref(st)))
            if i:
                ms.append(st.solve_ms)
        mm = med_min(ms)
        # SURVEY 8d, CC: per round 8(m+1) + 4 nnz + 4 nnz [comp gather] + 8 m; the Afforest passes of this solver each touch
        # at most that (sampling rounds read two neighbours per vertex, the link pass the rest): model x passes is an
        # upper bound of the algorithmic bytes, so `frac` is an upper bound too -- reported with the pass count
        b = (8 * (m + 1) + 8 * nnz + 8 * m) * max(st.iterations, 1)
        sec = mm["median"] * 1e-3
        rec[name] = {"ms": mm, "passes": st.iterations, "edges_per_s": nnz / sec,
                     "components": int((comp == torch.arange(m, dtype=torch.int32, device=device)).sum().item()),
                     "roofline": {"bound": "hbm", "achieved": b / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": b / sec / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": b,
                                  "frac_one_pass": (b // max(st.iterations, 1)) / sec / 1e9 / HBM_PEAK_GBS,
                                  "model": "(8(m+1) + 8 nnz + 8 m) x passes (SURVEY 8d, CC per round; upper bound for "
                                           "Afforest's sampling passes)"}}
        attach_traffic(rec[name]["roofline"], "cc" if rev is not None else "cc_out", args.trav_scale, sec)
    # out-edges only with the reverse graph built INSIDE the call (GDN_CC_REVERSE=build): its build lands in prep_ms, the solve is
    # the one with the giant-component skip -- wall time (prep + solve) is what the default avoids
    try:
        _cabi.check(L.gdn_option_set(b"GDN_CC_REVERSE", b"build"))
        shots = []
        for i in range(4):
            st = _cabi.GdnStats()
            _cabi.check(L.gdn_cc_dev(go, None, C.c_void_p(comp.data_ptr()), C.byref(st)))
            if i:
                shots.append({"solve_ms": st.solve_ms, "prep_ms": st.prep_ms})
        rec["cc_out_edges_only"]["option_reverse_built_in_call"] = {
            "shots": shots, "components": int((comp == torch.arange(m, dtype=torch.int32, device=device)).sum().item())}
    except Exception as e:
        log(f"[bench] cc with the reverse graph built in the call skipped: {e}")
    finally:
        L.gdn_option_set(b"GDN_CC_REVERSE", None)
    # the reference's fusion variant of CC (src/cc/fusion.cu:47: the Shiloach-Vishkin rounds inside one persistent kernel), for
    # the record: it sweeps all edges once per round where Afforest sweeps them once in all
    try:
        _cabi.check(L.gdn_option_set(b"GDN_CC_SV", b"fused"))
        ms = []
        for i in range(4):
            st = _cabi.GdnStats()
            _cabi.check(L.gdn_cc_dev(go, None, C.c_void_p(comp.data_ptr()), C.byref(st)))
            if i:
                ms.append(st.solve_ms)
        rec["cc_sv_fused_kernel"] = {"ms": med_min(ms), "rounds": st.iterations, "ran_fused": st.reserved == 2,
                                     "components": int((comp == torch.arange(m, dtype=torch.int32, device=device)).sum().item())}
    except Exception as e:
        log(f"[bench] fused SV skipped: {e}")
    finally:
        L.gdn_option_set(b"GDN_CC_SV", None)
    L.gdn_graph_free(go)
    L.gdn_graph_free(gi)
    log(f"[bench] traversal: {rec}")
    return rec
