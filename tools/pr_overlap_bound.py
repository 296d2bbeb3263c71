#!/usr/bin/env python3
"""What overlapping consecutive PageRank iterations could buy at most: TWO independent iteration chains (two plans of the same graph, own state
and contribution buffers) on two streams against the same two chains one after the other.  Inside one chain an iteration's phase A needs the
phase B in front of it, so its kernels run back to back and every kernel's tail -- the last round of workgroups -- leaves CUs idle; two chains
side by side fill those gaps with each other's work.  The gain of the pair over the sum is an UPPER bound of what pipelining phase A of
iteration i + 1 behind the bins of iteration i (tickets) could reach.  usage: pr_overlap_bound.py [scale 27] [steps 20]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gardenia_amd import _cabi, graphio

L = _cabi.lib()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.cuda.init()
go, gi = C.c_void_p(), C.c_void_p()
_cabi.check(L.gdn_rmat_build(scale, 16, graphio.K_RAND_SEED, 1, C.byref(go), C.byref(gi)))
m, nnz = C.c_int32(), C.c_uint64()
_cabi.check(L.gdn_graph_info(gi, C.byref(m), C.byref(nnz), None, None))
m, nnz = m.value, nnz.value


def alloc(nbytes):
    p = C.c_void_p()
    _cabi.check(L.gdn_dev_alloc(nbytes, C.byref(p)))
    return p


deg, scores0 = alloc(4 * m), alloc(4 * m)
_cabi.check(L.gdn_graph_degrees_dev(go, deg, None))
L.gdn_graph_free(go)
init = np.full(m, np.float32(1.0) / np.float32(m), np.float32)
_cabi.check(L.gdn_dev_upload(scores0, init.ctypes.data_as(C.c_void_p), 4 * m))
chains = []
for k in range(2):
    plan = C.c_void_p()
    _cabi.check(L.gdn_pr_plan_create(gi, deg, m, 0, 2, C.byref(plan)))
    ms_ = C.c_int32(0)
    _cabi.check(L.gdn_pr_plan_state_size(plan, C.byref(ms_)))
    state, c0, c1, diff = alloc(4 * ms_.value), alloc(4 * ms_.value), alloc(4 * ms_.value), alloc(8)
    _cabi.check(L.gdn_pr_import_dev(plan, scores0, state, 0.85, None))
    _cabi.check(L.gdn_pr_contrib_dev(plan, state, c0, None))
    chains.append(dict(plan=plan, state=state, bufs=[c0, c1], diff=diff, it=0, stream=torch.cuda.Stream()))


def pull(ch):
    s = C.c_void_p(ch["stream"].cuda_stream)
    _cabi.check(L.gdn_pr_pull_dev(ch["plan"], ch["bufs"][ch["it"] & 1], ch["state"], ch["bufs"][(ch["it"] + 1) & 1], ch["diff"], 0.85, s))
    ch["it"] += 1


def timed(which, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for k in which:
            pull(chains[k])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for k in (0, 1):
    timed([k], 5)
for rnd in range(3):
    a, b = timed([0], steps), timed([1], steps)
    both = timed([0, 1], steps)
    print("round %d: chain 0 alone %.3f ms/step, chain 1 alone %.3f, the two side by side %.3f ms per PAIR of steps = %.3f of the sum (%.1f %% gained)"
          % (rnd, a / steps, b / steps, both / steps, both / (a + b), 100.0 * (1.0 - both / (a + b))), flush=True)
