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
from gardenia_amd.sharded import ShardedPageRank, vertex_range  # noqa: E402


class NumpyBackend:
    """Row-range pull in numpy fp32 (test stand-in for HipPageRankBackend)."""

    def __init__(self, g_in, out_degree, m, lo, hi, chunk, world):
        self.m, self.lo, self.hi = m, lo, hi
        self.rowptr = g_in.rowptr[lo:hi + 1].astype(np.int64)
        self.colidx = g_in.colidx
        self.deg = out_degree[lo:hi].astype(np.float32)
        self.contribs = [torch.zeros(chunk * world, dtype=torch.float32) for _ in range(2)]
        self.scores = np.full(hi - lo, np.float32(1.0) / np.float32(m), np.float32)
        self.diff = torch.zeros(1, dtype=torch.float64)

    def contrib_full(self, which):
        return self.contribs[which]

    def diff_tensor(self):
        return self.diff

    def contrib(self, which):
        with np.errstate(divide="ignore"):
            self.contribs[which].numpy()[self.lo:self.hi] = self.scores / self.deg

    def pull(self, cin, cout, damping):
        c = self.contribs[cin].numpy()
        base = np.float32((np.float32(1.0) - np.float32(damping)) / np.float32(self.m))
        sums = np.zeros(self.hi - self.lo, np.float32)
        for r in range(self.hi - self.lo):
            acc = np.float32(0)
            for e in range(self.rowptr[r], self.rowptr[r + 1]):
                acc = np.float32(acc + c[self.colidx[e]])
            sums[r] = acc
        new = (base + np.float32(damping) * sums).astype(np.float32)
        self.diff[0] = float(np.abs((new - self.scores).astype(np.float32)).astype(np.float64).sum())
        self.scores = new
        with np.errstate(divide="ignore"):
            self.contribs[cout].numpy()[self.lo:self.hi] = new / self.deg


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
    lo, hi, chunk = vertex_range(rank, world, m)
    be = NumpyBackend(gi, g.degrees(), m, lo, hi, chunk, world)
    pr = ShardedPageRank(be, m, rank, world, dist)
    it, err = pr.solve()
    np.save(f"{out}.{rank}.npy", be.scores)
    if rank == 0:
        np.save(f"{out}.meta.npy", np.array([it, err]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
