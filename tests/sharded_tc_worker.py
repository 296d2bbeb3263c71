"""Worker for tests/test_sharded_cpu.py: one rank of a world_size-N gloo job that drives gardenia_amd.sharded.ShardedTC
with a TEST-SIDE numpy backend (the product backend needs a GPU).  Rank 0 writes the total to <out>."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gardenia_amd import graphio  # noqa: E402
from gardenia_amd.sharded import ShardedTC  # noqa: E402


class NumpyTCBackend:
    """Merge-intersect count over a row range of the DAG (test stand-in for HipTCBackend)."""

    def __init__(self, dag):
        self.dag, self.device = dag, torch.device("cpu")

    def count_rows(self, lo, hi):
        rp, ci = self.dag.rowptr.astype(np.int64), self.dag.colidx
        n = 0
        for u in range(lo, hi):
            nu = ci[rp[u]:rp[u + 1]]
            for v in nu:
                n += len(np.intersect1d(nu, ci[rp[v]:rp[v + 1]], assume_unique=True))
        return n


def main():
    scale, ef, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = graphio.symmetrize(graphio.rmat_graph(scale, ef, seed=78))
    dag = graphio.orient_dag(g)
    tc = ShardedTC(NumpyTCBackend(dag), dag.rowptr, rank, world, dist)
    total = tc.count()
    if rank == 0:
        np.save(out, np.array([total, tc.lo, tc.hi], dtype=np.int64))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
