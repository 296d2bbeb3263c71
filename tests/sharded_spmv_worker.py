"""Worker for tests/test_sharded_cpu.py: one rank of a gloo job driving gardenia_amd.sharded.ShardedSpMV with a
TEST-SIDE numpy backend (the product backend needs a GPU).  Writes its y slice to <out>.<rank>.npy."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gardenia_amd import graphio  # noqa: E402
from gardenia_amd.sharded import ShardedSpMV, vertex_range  # noqa: E402


class NumpyBackend:
    def __init__(self, g, Ax, x, y0, lo, hi, chunk, world):
        self.lo, self.hi = lo, hi
        self.rowptr = g.rowptr[lo:hi + 1].astype(np.int64)
        self.colidx, self.Ax = g.colidx, Ax
        self.x = torch.zeros(chunk * world, dtype=torch.float32)
        self.x[lo:hi] = torch.from_numpy(x[lo:hi].copy())  # only this rank's slice is known before the gather
        self.y = y0[lo:hi].copy()

    def x_full(self):
        return self.x

    def multiply(self):
        x = self.x.numpy()
        for r in range(self.hi - self.lo):
            acc = self.y[r]
            for e in range(self.rowptr[r], self.rowptr[r + 1]):
                acc = np.float32(acc + np.float32(self.Ax[e] * x[self.colidx[e]]))
            self.y[r] = acc


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = graphio.rmat_graph(8, 8, seed=78)
    m = g.m
    rng = np.random.default_rng(5)
    Ax = rng.random(g.nnz).astype(np.float32)
    x = rng.random(m).astype(np.float32)
    y0 = rng.random(m).astype(np.float32)
    lo, hi, chunk = vertex_range(rank, world, m)
    be = NumpyBackend(g, Ax, x, y0, lo, hi, chunk, world)
    sp = ShardedSpMV(be, m, rank, world, dist)
    sp.multiply()
    np.save(f"{out}.{rank}.npy", be.y)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
