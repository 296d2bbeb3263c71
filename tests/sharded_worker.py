"""Worker for tests/test_sharded_cpu.py: one rank of a world_size-N gloo job that drives
gardenia_amd.sharded.ShardedPageRank with a TEST-SIDE numpy backend (the product backend
needs a GPU).  Writes its score slice to <out>.<rank>.npy."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gardenia_amd import graphio  # noqa: E402
from gardenia_amd.sharded import (ShardedPageRank, edge_balanced_ranges, pad_columns, padded_chunk,  # noqa: E402
                                  vertex_range)


class NumpyBackend:
    """Row-range pull in numpy fp32 (test stand-in for HipPageRankBackend)."""

    def __init__(self, g_in, out_degree, m, lo, hi, chunk, world, row0=None, colidx=None):
        """Rows [row0, row0 + hi - lo) of g_in; they sit at [lo, hi) of the (possibly padded) vertex space that `colidx`
        (default: g_in's) addresses.  m = the ORIGINAL vertex count (base score)."""
        row0 = lo if row0 is None else row0
        self.m, self.lo, self.hi = m, lo, hi
        self.rowptr = g_in.rowptr[row0:row0 + (hi - lo) + 1].astype(np.int64)
        self.colidx = g_in.colidx if colidx is None else colidx
        self.deg = out_degree[row0:row0 + (hi - lo)].astype(np.float32)
        self.contribs = [torch.zeros(chunk * world + 4, dtype=torch.float32) for _ in range(2)]  # + the dummy slot
        self.scores = np.full(hi - lo, np.float32(1.0) / np.float32(m), np.float32)
        self.diff = torch.zeros(1, dtype=torch.float64)

    def contrib_full(self, which):
        return self.contribs[which]

    def active_sources(self):
        return torch.from_numpy(self.deg > 0)

    def diff_tensor(self):
        return self.diff

    def contrib(self, which):
        with np.errstate(divide="ignore"):
            self.contribs[which].numpy()[self.lo:self.hi] = self.scores / self.deg

    def pull(self, cin, cout, damping):
        self.pull_rows(cin, cout, damping, 0, self.hi - self.lo, True, True)

    def pull_rows(self, cin, cout, damping, r0, r1, first, last):
        """Rows [r0,r1) of this rank only (the contract of gdn_pr_pull_rows_dev: rows below r1 are final
        afterwards, the L1 change is complete after the last part)."""
        r0, r1 = min(r0, self.hi - self.lo), min(r1, self.hi - self.lo)
        c = self.contribs[cin].numpy()
        base = np.float32((np.float32(1.0) - np.float32(damping)) / np.float32(self.m))
        if first:
            self._acc = 0.0
        for r in range(r0, r1):
            acc = np.float32(0)
            for e in range(self.rowptr[r], self.rowptr[r + 1]):
                acc = np.float32(acc + c[self.colidx[e]])
            new = np.float32(base + np.float32(damping) * acc)
            self._acc += float(np.abs(np.float32(new - self.scores[r])))
            self.scores[r] = new
            with np.errstate(divide="ignore"):
                self.contribs[cout].numpy()[self.lo + r] = new / self.deg[r]
        if last:
            self.diff[0] = self._acc


class TicketedNumpyBackend(NumpyBackend):
    """The contract of gdn_pr_pull_parts_dev / gdn_pr_wait_part_dev on the CPU: pull_ticketed computes the WHOLE iteration at once
    (as the one launch per phase does) but publishes a part's rows only when part_ready(j) is entered -- until then the rows of
    that part hold a poison value in the vector the ranks exchange.  An exchange queued outside its part's context, or before
    it, would ship the poison and the scores would not equal the oracle's."""

    def pull_ticketed(self, cin, cout, damping, row_ends):
        ml = self.hi - self.lo
        self.pull_rows(cin, cout, damping, 0, ml, True, True)
        out = self.contribs[cout].numpy()
        self._held, self._ends, self._cout = out[self.lo:self.hi].copy(), [min(int(r), ml) for r in row_ends], cout
        out[self.lo:self.hi] = np.float32(-7.0)
        self._released = 0

    def part_ready(self, part):
        import contextlib
        assert part == self._released, "parts are released in order"
        r0 = self._ends[part - 1] if part else 0
        r1 = self._ends[part]
        self.contribs[self._cout].numpy()[self.lo + r0:self.lo + r1] = self._held[r0:r1]
        self._released += 1
        return contextlib.nullcontext()


def main():
    scale, ef, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = graphio.rmat_graph(scale, ef, seed=77)
    m = g.m - 3  # not divisible by the world size: exercises the padded all-gather
    keep = (graphio.csr_to_coo(g)[0] < m) & (graphio.csr_to_coo(g)[1] < m)
    src, dst = graphio.csr_to_coo(g)
    g = graphio.build_csr(m, src[keep], dst[keep])
    gi = graphio.transpose(g)
    Backend = TicketedNumpyBackend if os.environ.get("GDN_TEST_TICKETS") == "1" else NumpyBackend
    if os.environ.get("GDN_TEST_BALANCED") == "1":
        # nnz-balanced ranges in the padded vertex space (what bench.py --gpus N and gdn_pr_multi do)
        ranges = edge_balanced_ranges(gi.rowptr, world, min_rows=1)
        bounds = [a for a, _ in ranges] + [m]
        chunk = padded_chunk(bounds)
        blo, bhi = ranges[rank]
        lo, hi = rank * chunk, rank * chunk + (bhi - blo)
        be = Backend(gi, g.degrees(), m, lo, hi, chunk, world, row0=blo, colidx=pad_columns(gi.colidx, bounds, chunk))
        m_space = chunk * world
        if rank == 0:
            np.save(f"{out}.bounds.npy", np.array(bounds))
    else:
        lo, hi, chunk = vertex_range(rank, world, m)
        be = Backend(gi, g.degrees(), m, lo, hi, chunk, world)
        m_space = m
    parts = int(os.environ.get("GDN_TEST_PARTS", "4"))
    pr = ShardedPageRank(be, m_space, rank, world, dist, parts=parts, exchange=os.environ.get("GDN_TEST_EXCHANGE", "auto"))
    assert pr.exchange == os.environ.get("GDN_TEST_EXCHANGE", "dense"), pr.exchange
    if pr.exchange == "compact":
        assert pr.exchanged_bytes() < 4 * chunk * world
    it, err = pr.solve()
    np.save(f"{out}.{rank}.npy", be.scores)
    if rank == 0:
        np.save(f"{out}.meta.npy", np.array([it, err]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
