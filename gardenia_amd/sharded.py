"""Vertex-range sharded pull PageRank across the GPUs of one node.

The reference has no multi-GPU path; this is the MI355X-native addition the north star asks
for: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI), rows of the
in-CSR split into `world` contiguous vertex ranges, column ids global.  Ranges: equal vertex counts
(`vertex_range`) or -- what bench.py uses -- about nnz/world edges each (`gdn_graph_balanced_ranges`) with
the graph moved into a PADDED vertex space (`padded_chunk`, `gdn_graph_slice_padded`: range r occupies the slot
[r*chunk, r*chunk + len_r) of chunk*world ids), so that unequal ranges still exchange equal all-gather slots; in that
space every rank's range is again `vertex_range(rank, world, chunk*world)`, which is all this module needs to know.
Per iteration each rank

  1. runs the fused pull kernel on its rows: reads the full contrib vector, writes its slice of
     the next contrib vector and its local L1 change          (C-ABI gdn_pr_pull_dev)
  2. all-gathers the next contrib vector in place               (RCCL all-gather, m*4 B in all)
  3. all-reduces the 8-byte L1 change                           (convergence test)

Compact exchange (`exchange="compact"`, the default): only the contributions of vertices WITH out-edges are ever read by
anybody (39 % of RMAT-27's vertices), so a rank sends just those -- gathered into a dense send buffer, all-gathered,
scattered to their places in the full vector (index lists exchanged once at setup) -- 2.6x fewer bytes over xGMI.  xGMI is
point to point: a pair of GPUs shares ONE link (~77 GB/s per direction), so the dense exchange (N = 2: 256 MB over one link,
N = 8: 64 MB to each of 7 peers) takes longer than a rank's share of the compute at every N; the scatter costs 0.33 ms for
all of RMAT-27's 52 M active entries (measured, tools/scatter_probe.py), less the rank's own share.

Steps 1 and 2 are pipelined: the rank's rows are cut into `parts` row ranges; as soon as the kernel of part j has
been queued (gdn_pr_pull_rows_dev) its rows are all-gathered asynchronously (RCCL runs on its own stream, ordered
behind the compute stream at the call), so only the last part's exchange is exposed.  xGMI is point to point: the
64 MB a rank sends at N = 8 on RMAT-27 take longer than its share of the compute, so an un-overlapped exchange
would dominate the iteration.

The north star words step 2 as "all-reduce of the rank vector"; the all-gather moves half the
bytes for the same result (each slice has exactly one writer).  BFS/SSSP stay single-GPU.

`backend` abstracts the local kernel so that the orchestration (partition, buffer swap,
collectives, convergence) is exercised by world_size-2 gloo tests on CPU with a test-side
backend; the product backend is HipPageRankBackend (no CPU fallback inside this package).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple


def vertex_range(rank: int, world: int, m: int) -> Tuple[int, int, int]:
    """(lo, hi, chunk): rank owns rows [lo,hi); chunk = ceil(m/world) is the all-gather unit."""
    chunk = (m + world - 1) // world
    lo = min(rank * chunk, m)
    hi = min(lo + chunk, m)
    return lo, hi, chunk


def padded_chunk(bounds) -> int:
    """Slot length of the padded vertex space for the ranges [bounds[r], bounds[r+1]): the longest range, rounded up to
    4 entries (16-byte aligned slices)."""
    longest = max(int(bounds[r + 1]) - int(bounds[r]) for r in range(len(bounds) - 1))
    return (longest + 3) & ~3


class ShardedPageRank:
    def __init__(self, backend, m_global: int, rank: int = 0, world: int = 1, dist=None,
                 damping: float = 0.85, parts: int = 4, exchange: str = "auto", first_diff_extra: float = 0.0,
                 force_collectives: bool = False):
        """force_collectives (test aid): run the exchange collectives even with a single rank.  first_diff_extra: L1 change of vertices OUTSIDE the sharded state in the first iteration of solve() (the
        dead vertices of a squished graph move from 1/m to the base score; gdn_pr_squish_import_dev reports it)."""
        self.be = backend
        self._multi = world > 1 or (force_collectives and dist is not None)
        self.parts = max(1, int(parts)) if self._multi and hasattr(backend, "pull_rows") else 1
        self.m = m_global
        self.rank, self.world, self.dist = rank, world, dist
        self.lo, self.hi, self.chunk = vertex_range(rank, world, m_global)
        self.damping = damping
        self.first_diff_extra = float(first_diff_extra)
        self.cur = 0  # index of the contrib buffer holding the current iteration's input
        self.iterations = 0
        self.n_full = self.chunk * world
        self._diff_cache = None
        self._stage = {}          # (r0, r1) -> staging buffer of a part's all-gather
        self._copy_stream = None  # where the strided copies out of the staging buffers run
        # Which collectives this backend takes is decided ONCE, here, on dummy tensors, and agreed by all ranks
        # (all-reduce MIN): no iteration is ever repeated, and no rank can take another path than its peers.
        self._inplace, list_ok = (True, True) if (not self._multi or dist is None) else self._probe_collectives()
        if not list_ok:
            self.parts = 1  # the pipelined exchange gathers into strided views
        # exchange: "dense" = every row's contribution, "compact" = only rows with out-edges (see the module docstring)
        self.exchange = "dense"
        self._cx = None
        if self._multi and dist is not None and hasattr(backend, "active_sources") and exchange in ("compact", "auto"):
            ok = 1
            try:
                if self.be.contrib_full(0).numel() < self.n_full + 1:
                    raise ValueError("the contrib buffers have no dummy slot behind chunk * world entries")
            except (RuntimeError, ValueError, NotImplementedError) as e:
                import sys
                print(f"[sharded] compact exchange unavailable ({e}); using the dense all-gather", file=sys.stderr, flush=True)
                ok = 0
            if self._agree(ok):
                self._setup_compact()
                self.exchange = "compact"

    def _agree(self, ok: int) -> bool:
        """True iff EVERY rank says ok (all-reduce MIN)."""
        import torch
        t = torch.tensor([int(ok)], dtype=torch.int32, device=self.be.contrib_full(0).device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def _probe_collectives(self):
        """(in-place all_gather_into_tensor ok, all_gather into a list of strided views ok) for this process group, tried
        on dummy tensors of the contrib vector's device; a form ANY rank cannot run is off for all of them."""
        import torch
        dev = self.be.contrib_full(0).device
        flags = []
        for form in ("inplace", "list"):
            ok = 1
            try:
                buf = torch.zeros(8 * self.world, dtype=torch.float32, device=dev)
                if form == "inplace":
                    self.dist.all_gather_into_tensor(buf, buf[8 * self.rank:8 * self.rank + 8])
                else:
                    outs = [buf[8 * r + 4:8 * r + 8] for r in range(self.world)]
                    self.dist.all_gather(outs, outs[self.rank], async_op=True).wait()
            except (RuntimeError, ValueError, NotImplementedError, TypeError) as e:
                import sys
                print(f"[sharded] rank {self.rank}: {form} all-gather unavailable on this backend ({e})", file=sys.stderr, flush=True)
                ok = 0
            flags.append(ok)
        t = torch.tensor(flags, dtype=torch.int32, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(int(t[0].item())), bool(int(t[1].item()))

    def _setup_compact(self):
        """Index lists of the compact exchange, one set per pipeline part: `my` = positions (in the full vector) of this
        rank's rows with out-edges inside the part, padded to the largest count over the ranks with the dummy slot
        n_full; `all` = the same lists of every rank, in all-gather order."""
        import torch
        full = self.be.contrib_full(0)
        act = self.be.active_sources()
        ml = self.hi - self.lo
        cx = []
        for (r0, r1) in (self.part_ranges() if self.parts > 1 else [(0, self.chunk)]):
            a0, a1 = min(r0, ml), min(r1, ml)
            idx = torch.nonzero(act[a0:a1]).flatten().to(torch.int64) + a0
            n = int(idx.numel())
            cap_t = torch.tensor([n], dtype=torch.int64, device=full.device)
            self.dist.all_reduce(cap_t, op=self.dist.ReduceOp.MAX)
            cap = max(int(cap_t.item()), 1)
            my = torch.full((cap,), self.n_full, dtype=torch.int64, device=full.device)
            my[:n] = idx + self.rank * self.chunk
            allg = torch.empty(self.world * cap, dtype=torch.int64, device=full.device)
            self.dist.all_gather_into_tensor(allg, my)
            cx.append({"my": my, "all": allg, "recv": torch.empty(self.world * cap, dtype=full.dtype, device=full.device)})
        self._cx = cx

    def exchanged_bytes(self) -> int:
        """Bytes one rank receives per iteration (incl. its own slice)."""
        if self._cx is None:
            return 4 * self.n_full
        return sum(4 * int(c["recv"].numel()) for c in self._cx)

    def init_contrib(self):
        """contrib = score/out_degree for the local rows, then gather (src/pr/base.cu:14)."""
        self.be.contrib(self.cur)
        self._gather(self.cur)

    def _gather(self, which):
        if self._multi:
            full = self.be.contrib_full(which)[:self.n_full]
            mine = full[self.rank * self.chunk:(self.rank + 1) * self.chunk]
            # in-place: each rank's slice already sits at its place in `full` (a backend that rejects the aliasing --
            # found by the probe at setup -- gathers from a copy)
            self.dist.all_gather_into_tensor(full, mine if self._inplace else mine.clone())

    def part_ranges(self):
        """Row ranges [r0,r1) (relative to a rank's first row) of the pipeline parts: equal on every rank,
        multiples of 4 rows (16-byte aligned slices), the last part takes the remainder of the chunk."""
        seg = -(-self.chunk // self.parts)
        seg = (seg + 3) & ~3
        out = []
        for j in range(self.parts):
            r0 = min(j * seg, self.chunk)
            r1 = self.chunk if j == self.parts - 1 else min((j + 1) * seg, self.chunk)
            if r1 > r0 or j == 0:
                out.append((r0, r1))
        return out

    def _gather_rows_async(self, which, r0, r1):
        """All-gather rows [r0,r1) of every rank's slice; returns an object whose wait() makes the CURRENT stream wait for the
        rows of all ranks to be in place.  The rows are strided in the full vector (rank r's at r * chunk + [r0,r1)), and a
        list of strided views is gathered by torch.distributed through a flat buffer of its own plus one copy PER RANK.  So the
        part goes through ONE contiguous all_gather_into_tensor into a staging buffer of this driver (world x (r1 - r0), kept
        per part) and ONE strided copy back -- on a copy stream that waits for the collective, beside the accumulation of the
        later parts (8 ranks, 4 parts: 4 + 4 launches on the side streams per iteration instead of 4 + 32).  With ONE rank the
        part is contiguous and goes through the in-place form, which moves nothing."""
        full = self.be.contrib_full(which)
        if self.world == 1 and self._inplace:
            return self.dist.all_gather_into_tensor(full[r0:r1], full[r0:r1], async_op=True)
        n = r1 - r0
        key = (r0, r1)
        if key not in self._stage:
            self._stage[key] = full.new_empty(self.world * n)
        stage = self._stage[key]
        send = full[self.rank * self.chunk + r0:self.rank * self.chunk + r1]
        work = self.dist.all_gather_into_tensor(stage, send, async_op=True)
        return _StagedGather(self, work, stage, full[:self.n_full].view(self.world, self.chunk)[:, r0:r1], n)

    def _pull_part(self, nxt, ranges, j):
        """Queue the computation of pipeline part j and return the context in which its exchange is to be queued.
        A backend with tickets (HipPageRankBackend: gdn_pr_pull_parts_dev) runs the whole iteration as ONE launch per phase
        when part 0 is asked for -- its bins in part order -- and `part_ready(j)` is a side stream on which a one-wave kernel
        waits for part j's tickets; other backends (the CPU test backends) compute part j now, on the current stream."""
        import contextlib
        r0, r1 = ranges[j]
        if len(ranges) > 1 and hasattr(self.be, "pull_ticketed"):
            if j == 0:
                self.be.pull_ticketed(self.cur, nxt, self.damping, [r[1] for r in ranges])
            return self.be.part_ready(j)
        if len(ranges) > 1:
            self.be.pull_rows(self.cur, nxt, self.damping, r0, r1, first=(j == 0), last=(j == len(ranges) - 1))
        else:
            self.be.pull(self.cur, nxt, self.damping)
        return contextlib.nullcontext()

    def _step_compact(self, nxt):
        full = self.be.contrib_full(nxt)
        ranges = self.part_ranges() if self.parts > 1 else [(0, self.chunk)]
        works = []
        for j, (r0, r1) in enumerate(ranges):
            with self._pull_part(nxt, ranges, j):
                cx = self._cx[j]
                send = full.index_select(0, cx["my"])  # pad entries read the dummy slot
                works.append((self.dist.all_gather_into_tensor(cx["recv"], send, async_op=True), send, cx))
        for w, _send, cx in works:
            w.wait()
            cap = cx["my"].numel()
            a, b = self.rank * cap, (self.rank + 1) * cap  # this rank's own entries are already in place
            if a > 0:
                full.index_copy_(0, cx["all"][:a], cx["recv"][:a])  # pad entries land in the dummy slot
            if b < cx["all"].numel():
                full.index_copy_(0, cx["all"][b:], cx["recv"][b:])

    def step(self):
        """One PageRank iteration; returns nothing (the L1 change stays on the device).  The exchange form was fixed at
        setup (probe + agreement of all ranks): a collective that fails here is an error, never a silent redo -- the pull
        updates the scores in place, so a repeated pull would report an L1 change of ~0 and fake convergence."""
        nxt = self.cur ^ 1
        if self._cx is not None:
            self._step_compact(nxt)
        elif self.parts <= 1:
            self.be.pull(self.cur, nxt, self.damping)
            self._gather(nxt)
        else:
            ranges = self.part_ranges()
            works = []
            for j, (r0, r1) in enumerate(ranges):
                with self._pull_part(nxt, ranges, j):
                    if r1 > r0:
                        works.append(self._gather_rows_async(nxt, r0, r1))
            for w in works:
                w.wait()
        self.cur = nxt
        self.iterations += 1
        self._diff_cache = None

    def global_diff(self) -> float:
        """Sum of the local L1 changes of the LAST step over all ranks (blocking; safe to call more than once)."""
        if self._diff_cache is None:
            d = self.be.diff_tensor().clone()  # never reduce the backend's tensor in place: a second call would
            if self._multi:                    # return world times the value
                self.dist.all_reduce(d)
            self._diff_cache = float(d.item())
        return self._diff_cache

    def solve(self, epsilon: float = 1e-4, max_iter: int = 100) -> Tuple[int, float]:
        """Iterate like src/pr/omp_base.cc:20-37; returns (iterations as printed, last error)."""
        self.init_contrib()
        err = 0.0
        it = 0
        extra = self.first_diff_extra
        if hasattr(self.be, "first_iteration_extra"):
            extra += float(self.be.first_iteration_extra())
        for it in range(max_iter):
            self.step()
            err = self.global_diff()
            if it == 0:
                err += extra  # vertices outside the state (no edge at all) move to the base score in iteration 1
            if err < epsilon:
                break
        return it + 1, err


class _StagedGather:
    """A part's all-gather into a staging buffer + the strided copy into the full vector (ShardedPageRank._gather_rows_async)."""

    def __init__(self, owner, work, stage, dst, n):
        self.work, self.stage, self.dst, self.n, self.ev = work, stage, dst, n, None
        torch = None
        if stage.is_cuda:
            import torch
        if torch is not None:
            if owner._copy_stream is None:
                owner._copy_stream = torch.cuda.Stream(device=stage.device)
            with torch.cuda.stream(owner._copy_stream):
                self.work.wait()  # the copy stream waits for the collective ...
                self.dst.copy_(self.stage.view(-1, n))  # ... and puts every rank's rows in their place
                self.ev = torch.cuda.Event()
                self.ev.record(owner._copy_stream)
            self.torch = torch

    def wait(self):
        if self.ev is not None:
            self.torch.cuda.current_stream().wait_event(self.ev)
        else:  # CPU tensors (the gloo tests): the collective, then the copy, here
            self.work.wait()
            self.dst.copy_(self.stage.view(-1, self.n))


class _Entered:
    """A context manager that has been entered already (HipPageRankBackend.part_ready): `with` only leaves it."""

    def __init__(self, ctx):
        self._ctx = ctx

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return self._ctx.__exit__(*exc)


class HipPageRankBackend:
    """Local shard on one MI355X: torch device tensors + the _dev entry points of the C-ABI."""

    def __init__(self, torch, in_csr_handle, out_degree_local, m_global: int, lo: int, hi: int, chunk: int,
                 world: int, device, layout: int = -1, m_base: int = 0, force_sharded: bool = False):
        """m_global: size of the vertex space the shard's column ids live in (the padded space chunk * world of a sharded
        run); m_base > 0: the ORIGINAL vertex count, which sets the base score and the start scores (a squished and / or
        padded space has another size).  force_sharded: treat a single rank like one of many (no plan-internal squish)."""
        from . import _cabi
        self.torch, self._cabi, self.L = torch, _cabi, _cabi.lib()
        self.device = device
        self.m_local = hi - lo
        self.out_degree = out_degree_local  # int32 device tensor, m_local
        self.m_global = m_global
        n_full = chunk * world
        self.diff = torch.zeros(1, dtype=torch.float64, device=device)
        self._side = None  # the stream the part waiters and the exchanges behind them are queued on (part_ready)
        self.plan = C.c_void_p()
        # a single rank that holds the whole graph works on the LIVE vertices only (GDN_LAYOUT_PB_SQUISHED: vertices without
        # any edge keep the base score and are left out of the per-iteration state; GDN_PR_SQUISH=0 keeps them in)
        import os
        single = world == 1 and not force_sharded
        if single and layout in (_cabi.GDN_LAYOUT_AUTO, _cabi.GDN_LAYOUT_PB) and os.environ.get("GDN_PR_SQUISH", "1") != "0":
            nnz = C.c_uint64(0)
            _cabi.check(self.L.gdn_graph_info(in_csr_handle, None, C.byref(nnz), None, None))
            env = os.environ.get("GDN_PR_LAYOUT", "")
            if layout == _cabi.GDN_LAYOUT_PB or env[:1] == "p" or (env[:1] != "c" and nnz.value >= (1 << 22)):
                layout = _cabi.GDN_LAYOUT_PB_SQUISHED
        _cabi.check(self.L.gdn_pr_plan_create(in_csr_handle, C.c_void_p(out_degree_local.data_ptr()), m_global, lo,
                                              layout, C.byref(self.plan)))
        if m_base:
            _cabi.check(self.L.gdn_pr_plan_set_base(self.plan, int(m_base)))
        lay, lg, ms = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _cabi.check(self.L.gdn_pr_plan_layout(self.plan, C.byref(lay), C.byref(lg)))
        _cabi.check(self.L.gdn_pr_plan_state_size(self.plan, C.byref(ms)))
        self.layout, self.log_blk, self.m_state = lay.value, lg.value, ms.value
        self.squished = single and self.m_state != self.m_local
        n_vec = self.m_state if self.squished else n_full
        # + 4 entries behind the vector: [n_full] is the dummy slot of the compact exchange (16-byte alignment kept)
        self.contribs = [torch.zeros(n_vec + 4, dtype=torch.float32, device=device) for _ in range(2)]
        full = torch.full((max(self.m_local, 1),), 1.0 / (m_base or m_global), dtype=torch.float32, device=device)
        if self.squished:
            self.scores = torch.empty(max(self.m_state, 1), dtype=torch.float32, device=device)
            _cabi.check(self.L.gdn_pr_import_dev(self.plan, C.c_void_p(full.data_ptr()), C.c_void_p(self.scores.data_ptr()),
                                                 0.85, self._stream()))
            torch.cuda.synchronize()
            del full
        else:
            self.scores = full

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def contrib_full(self, which):
        return self.contribs[which]

    def first_iteration_extra(self) -> float:
        """L1 change of the vertices outside this plan's state in the first iteration after the import (squished plan)."""
        d = C.c_double(0.0)
        self._cabi.check(self.L.gdn_pr_import_diff(self.plan, C.byref(d)))
        return d.value

    def active_sources(self):
        """Rows of this rank whose contribution anybody reads: the vertices with out-edges."""
        return self.out_degree > 0

    def diff_tensor(self):
        return self.diff

    def contrib(self, which):
        self._cabi.check(self.L.gdn_pr_contrib_dev(self.plan, C.c_void_p(self.scores.data_ptr()),
                                                   C.c_void_p(self.contribs[which].data_ptr()), self._stream()))

    def pull(self, cin, cout, damping):
        self._cabi.check(self.L.gdn_pr_pull_dev(self.plan, C.c_void_p(self.contribs[cin].data_ptr()),
                                                C.c_void_p(self.scores.data_ptr()),
                                                C.c_void_p(self.contribs[cout].data_ptr()),
                                                C.c_void_p(self.diff.data_ptr()), float(damping), self._stream()))

    def pull_rows(self, cin, cout, damping, r0, r1, first, last):
        """Rows [r0,r1) of this rank (clipped to its row count) of one iteration: gdn_pr_pull_rows_dev."""
        flags = (self._cabi.GDN_PR_PART_FIRST if first else 0) | (self._cabi.GDN_PR_PART_LAST if last else 0)
        r0c, r1c = min(r0, self.m_local), min(r1, self.m_local)
        self._cabi.check(self.L.gdn_pr_pull_rows_dev(self.plan, C.c_void_p(self.contribs[cin].data_ptr()),
                                                     C.c_void_p(self.scores.data_ptr()),
                                                     C.c_void_p(self.contribs[cout].data_ptr()),
                                                     C.c_void_p(self.diff.data_ptr()), float(damping), r0c, r1c, flags,
                                                     self._stream()))

    def pull_ticketed(self, cin, cout, damping, row_ends):
        """One iteration as ONE launch per phase whose rows become final part by part (gdn_pr_pull_parts_dev): row_ends[j]
        ends part j (clipped to this rank's row count)."""
        n = len(row_ends)
        ends = (C.c_int32 * n)(*[min(int(r), self.m_local) for r in row_ends])
        self._cabi.check(self.L.gdn_pr_pull_parts_dev(self.plan, C.c_void_p(self.contribs[cin].data_ptr()),
                                                      C.c_void_p(self.scores.data_ptr()),
                                                      C.c_void_p(self.contribs[cout].data_ptr()),
                                                      C.c_void_p(self.diff.data_ptr()), float(damping), n, ends, self._stream()))

    def part_ready(self, part: int):
        """Context manager: inside it the current stream is this backend's side stream, on which a one-wave kernel has been
        queued that ends when part `part` of the last pull_ticketed is final (gdn_pr_wait_part_dev) -- what is queued inside
        (the exchange of the part) runs beside the accumulation of the later parts."""
        if self._side is None:
            self._side = self.torch.cuda.Stream(device=self.device)
        ctx = self.torch.cuda.stream(self._side)
        ctx.__enter__()
        try:
            self._cabi.check(self.L.gdn_pr_wait_part_dev(self.plan, int(part), C.c_void_p(self._side.cuda_stream)))
        except BaseException:
            ctx.__exit__(None, None, None)
            raise
        return _Entered(ctx)

    def n_bins(self) -> int:
        """Workgroups of the accumulate phase on this rank (0: CSR layout)."""
        nb = C.c_int32(0)
        self._cabi.check(self.L.gdn_pr_plan_bins(self.plan, C.byref(nb)))
        return nb.value

    def export_scores(self, damping: float = 0.85):
        """The rank's scores in the caller's vertex order (m_local entries)."""
        if not self.squished:
            return self.scores
        out = self.torch.empty(self.m_local, dtype=self.torch.float32, device=self.device)
        self._cabi.check(self.L.gdn_pr_export_dev(self.plan, C.c_void_p(self.scores.data_ptr()), C.c_void_p(out.data_ptr()),
                                                  float(damping), self._stream()))
        return out

    def check(self):
        """Raise if the PB fixed-point accumulator saw an out-of-range value (blocking)."""
        self._cabi.check(self.L.gdn_pr_plan_check(self.plan))

    def iter_bytes(self) -> int:
        return int(self.L.gdn_pr_iter_bytes(self.plan))

    def arm_kernel_timing(self, max_launches: int):
        self._cabi.check(self.L.gdn_pr_plan_kernel_time(self.plan, 1, max_launches, None, None))

    def read_kernel_timing(self):
        tot, n = (C.c_double * 2)(0, 0), C.c_int32(0)
        self._cabi.check(self.L.gdn_pr_plan_kernel_time(self.plan, 0, 0, tot, C.byref(n)))
        return (tot[0], tot[1]), n.value

    def close(self):
        if self.plan:
            self.L.gdn_pr_plan_free(self.plan)
            self.plan = C.c_void_p()


class ShardedSpMV:
    """y = y + A x with the rows of A split into vertex ranges like ShardedPageRank: every rank holds
    rows [lo,hi) with global column ids and its slice of x; one all-gather of x (m*4 B) precedes the local multiply,
    y stays distributed.  Iterated use (power method) repeats exactly this exchange per multiply."""

    def __init__(self, backend, m_global: int, rank: int = 0, world: int = 1, dist=None, inplace: bool = False):
        """inplace: gather x in place (RCCL and gloo take the aliasing; off by default: one copy of the slice)."""
        self.be, self.m, self.rank, self.world, self.dist = backend, m_global, rank, world, dist
        self.lo, self.hi, self.chunk = vertex_range(rank, world, m_global)
        self.inplace = inplace

    def gather_x(self):
        if self.world > 1:
            full = self.be.x_full()
            mine = full[self.rank * self.chunk:(self.rank + 1) * self.chunk]
            self.dist.all_gather_into_tensor(full, mine if self.inplace else mine.clone())

    def multiply(self):
        self.gather_x()
        self.be.multiply()


class HipSpMVBackend:
    """Local row shard on one MI355X: plan over rows [lo,hi) x all columns (gdn_spmv_plan_create_cols)."""

    def __init__(self, torch, shard_handle, Ax_local, m_global: int, lo: int, hi: int, chunk: int, world: int, device,
                 layout: int = -1):
        from . import _cabi
        self.torch, self._cabi, self.L = torch, _cabi, _cabi.lib()
        self.m_local = hi - lo
        self.Ax = Ax_local  # float32 device tensor, nnz of the shard, CSR order
        self.x = torch.zeros(chunk * world, dtype=torch.float32, device=device)
        self.y = torch.zeros(max(self.m_local, 1), dtype=torch.float32, device=device)
        self.plan = C.c_void_p()
        _cabi.check(self.L.gdn_spmv_plan_create_cols(shard_handle, C.c_void_p(Ax_local.data_ptr()), m_global, layout,
                                                      C.byref(self.plan)))

    def x_full(self):
        return self.x

    def multiply(self):
        s = C.c_void_p(self.torch.cuda.current_stream().cuda_stream)
        self._cabi.check(self.L.gdn_spmv_dev(self.plan, C.c_void_p(self.Ax.data_ptr()), C.c_void_p(self.x.data_ptr()),
                                             C.c_void_p(self.y.data_ptr()), s))

    def close(self):
        if self.plan:
            self._cabi.check(self.L.gdn_spmv_plan_check(self.plan))
            self.L.gdn_spmv_plan_free(self.plan)
            self.plan = C.c_void_p()


def edge_balanced_ranges(rowptr, world: int, min_rows: int = 0):
    """Row ranges [lo,hi) of `world` ranks with about nnz/world edges each (binary search on the row offsets).
    rowptr: numpy array of m+1 offsets.  min_rows = 1: every range keeps at least one row (what the C-ABI's
    gdn_graph_balanced_ranges / gdn_multi_ranges do -- a PageRank / SpMV shard without rows has no plan)."""
    import numpy as np
    rp = np.asarray(rowptr).astype(np.int64)
    m, nnz = len(rp) - 1, int(rp[-1])
    assert min_rows * world <= m
    bounds = [0]
    for r in range(1, world):
        b = int(np.searchsorted(rp, nnz * r // world, side="left"))
        b = max(b, bounds[-1] + min_rows)
        bounds.append(min(b, m - min_rows * (world - r)))
    bounds.append(m)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def pad_columns(colidx, bounds, chunk: int):
    """Column ids moved into the padded vertex space (numpy mirror of gdn_graph_slice_padded's relabelling):
    vertex v of range r -> r * chunk + (v - bounds[r])."""
    import numpy as np
    b = np.asarray(bounds, dtype=np.int64)
    c = np.asarray(colidx).astype(np.int64)
    r = np.searchsorted(b, c, side="right") - 1
    return (r * chunk + (c - b[r])).astype(np.int32)


class ShardedTC:
    """Triangle count over `world` ranks: every rank holds the oriented graph, counts the
    triangles whose lowest-ranked vertex lies in its row range -- ranges of equal DAG-edge count -- and ONE 8-byte
    all-reduce(sum) gives the total.  No other exchange: the count reads the whole DAG but writes nothing shared."""

    def __init__(self, backend, dag_rowptr, rank: int = 0, world: int = 1, dist=None):
        self.be, self.rank, self.world, self.dist = backend, rank, world, dist
        self.ranges = edge_balanced_ranges(dag_rowptr, world)
        self.lo, self.hi = self.ranges[rank]

    def count(self) -> int:
        local = int(self.be.count_rows(self.lo, self.hi))
        if self.world <= 1:
            return local
        import torch
        t = torch.tensor([local], dtype=torch.int64, device=self.be.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())


class HipTCBackend:
    """The oriented graph resident on one MI355X (gdn_graph_orient unless already a DAG) + gdn_tc_rows_dev."""

    def __init__(self, graph_handle, oriented: bool, device):
        from . import _cabi
        self._cabi, self.L, self.device = _cabi, _cabi.lib(), device
        self.own = C.c_void_p()
        self.dag = graph_handle
        if not oriented:
            _cabi.check(self.L.gdn_graph_orient(graph_handle, C.byref(self.own)))
            self.dag = self.own
        self.last_ms = 0.0

    def rowptr(self):
        import numpy as np
        m, nnz, rp = C.c_int32(), C.c_uint64(), C.c_void_p()
        self._cabi.check(self.L.gdn_graph_info(self.dag, C.byref(m), C.byref(nnz), C.byref(rp), None))
        h = np.empty(m.value + 1, np.uint64)
        self._cabi.check(self.L.gdn_dev_download(h.ctypes.data_as(C.c_void_p), rp, 8 * (m.value + 1)))
        return h

    def count_rows(self, lo: int, hi: int) -> int:
        total, st = C.c_uint64(0), self._cabi.GdnStats()
        self._cabi.check(self.L.gdn_tc_rows_dev(self.dag, lo, hi, C.byref(total), C.byref(st)))
        self.last_ms = st.solve_ms
        return int(total.value)

    def close(self):
        if self.own:
            self.L.gdn_graph_free(self.own)
            self.own = C.c_void_p()
