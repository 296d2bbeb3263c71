#!/usr/bin/env python3
"""Speed of plain device writes / reads / a strided scatter by WHERE in the device memory a buffer lies: the whole free memory is
taken in chunks of `chunk_gb` (allocation order = the order hipMalloc hands physical memory out in a fresh process), every chunk is
timed with three kernels (torch: fill_, a sum, an index_add_-free strided copy), median of `reps`.  The question behind it: PageRank's
phase A runs at three speeds depending on the physical pages of one 8.5 GB array (profiles/r05_pb_place_offsets.txt) -- do plain
streams see regions too?   usage: hbm_region_map.py [chunk_gb] [reps] [keep_free_gb]"""
import sys
import time

import torch

chunk_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
keep_free = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
dev = torch.device("cuda", 0)
n = int(chunk_gb * (1 << 30)) // 4
free, total = torch.cuda.mem_get_info()
print("free %.1f GB of %.1f GB; chunks of %.1f GB" % (free / 2**30, total / 2**30, chunk_gb), flush=True)
chunks = []
while True:
    free, _ = torch.cuda.mem_get_info()
    if free < (keep_free + chunk_gb) * 2**30:
        break
    try:
        chunks.append(torch.empty(n, dtype=torch.float32, device=dev))
    except RuntimeError:
        break
print("%d chunks allocated" % len(chunks), flush=True)


def timed(fn):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


# a scatter like phase A's: 4096 "bins" written side by side -- element i of the source goes to bin (i mod 4096), position i / 4096
src = torch.rand(n, dtype=torch.float32, device=dev)
bins = 4096
print("chunk  address            write GB/s   read GB/s   scatter GB/s (view-transposed copy, %d streams)" % bins)
for i, c in enumerate(chunks):
    w = timed(lambda: c.fill_(1.0))
    r = timed(lambda: c.sum())
    s = timed(lambda: c.view(bins, n // bins).copy_(src.view(n // bins, bins).t()))
    gb = n * 4 / 1e9
    print("%4d   0x%x   %9.0f   %9.0f   %9.0f" % (i, c.data_ptr(), gb / (w * 1e-3), gb / (r * 1e-3), 2 * gb / (s * 1e-3)), flush=True)
