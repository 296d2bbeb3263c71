// gdn_expand.hpp -- load-balanced neighbour expansion of a vertex list for wave64.
//
// Supersedes the reference's three-tier expand (src/bfs/linear_lb.cu:24-182: expandByCta /
// expandByWarp / CUB BlockScan + 256-entry LDS gather scratch; the same in
// src/sssp/linear_lb.cu:24-167), which hard-codes 32-lane warps and CUB:
//
//   * a wavefront takes 64 list entries (one per lane);
//   * rows with degree >= EXP_BIG are cut into EXP_CHUNK-edge work items pushed to a device
//     list (wave-aggregated push) and consumed by expand_big_kernel, so one hub never
//     serialises on one wave (RMAT-27 hubs have ~10^5..10^6 neighbours);
//   * rows with 64 <= degree < EXP_BIG are walked by the whole wave, one row at a time,
//     column_indices read as consecutive dwords (coalesced 256-B wave loads);
//   * rows with degree < 64 are packed: wave prefix sum of the degrees (shuffles), then each
//     step hands 64 consecutive packed edges to the lanes, owner found by a 6-step binary
//     search over the per-wave LDS copy of the prefix sums.
//
// The visitor is called CONVERGENTLY by all 64 lanes:
//     vis.edge(owner_lane, k, valid)
// owner_lane = lane that holds the source vertex' per-lane state (fetch it with __shfl),
// k = edge index (64-bit), valid = this lane has an edge.  Visitors may therefore use the
// wave-aggregated worklist push of gdn_common.hpp.
#pragma once
#include "gdn_common.hpp"

#define EXP_BIG 512      // rows at least this long go to the big-row list (a small frontier of medium rows would otherwise sit on a handful of waves)
#define EXP_CHUNK 256    // edges per big-row work item

struct ExpBigList {
  unsigned long long *items;  // (chunk << 32) | vertex
  unsigned *count;
  unsigned capacity;
  unsigned *overflow;
  unsigned min_deg = EXP_BIG;  // rows at least this long become work items (>= 64; lower it for SMALL vertex lists,
                               // whose medium rows would otherwise be walked one at a time by a handful of waves)
};

#ifdef __HIPCC__

// Expand the rows held one-per-lane: [b,e) is this lane's edge range (b == e for idle lanes),
// v its vertex (only used for the big list).  s_scan: 64 unsigned per wave of LDS.
template <class V>
__device__ __forceinline__ void gdn_expand_wave(eoff_t b, eoff_t e, vid_t v, ExpBigList big, V &vis,
                                                unsigned *s_scan) {
  const unsigned lane = gdn_lane();
  unsigned deg = (unsigned)((e - b) > 0xFFFFFFFFull ? 0xFFFFFFFFull : (e - b));

  // ---- tier 1: big rows -> chunk work items
  if (big.items != nullptr) {
    const bool is_big = deg >= big.min_deg;
    const unsigned nchunks = is_big ? (unsigned)((e - b + EXP_CHUNK - 1) / EXP_CHUNK) : 0u;
    const unsigned incl = gdn_wave_incl_scan(nchunks);
    const unsigned total = __shfl(incl, 63, 64);
    if (total) {
      unsigned base = 0;
      if (lane == 63) base = atomicAdd(big.count, total);
      base = __shfl(base, 63, 64);
      const unsigned mine = base + incl - nchunks;
      // the whole wave writes the items of one big row at a time: a hub of 10^6 edges is thousands of items, and
      // one lane writing them alone held a 456-vertex BFS level for 0.35 ms
      unsigned long long bigmask = __ballot(nchunks > 0u);
      while (bigmask) {
        const int leader = __ffsll((long long)bigmask) - 1;
        bigmask &= bigmask - 1ull;
        const unsigned n = __shfl(nchunks, leader, 64);
        const unsigned first = __shfl(mine, leader, 64);
        const unsigned vv = (unsigned)__shfl(v, leader, 64);
        for (unsigned c = lane; c < n; c += 64) {
          if (first + c < big.capacity) big.items[first + c] = ((unsigned long long)c << 32) | vv;
          else *big.overflow = 1u;
        }
      }
    }
    if (is_big) deg = 0;
  }

  // ---- tier 2: medium rows, whole wave per row
  {
    unsigned long long mask = __ballot(deg >= 64u);
    while (mask) {
      const int leader = __ffsll((long long)mask) - 1;
      mask &= mask - 1ull;
      const eoff_t bb = __shfl(b, leader, 64);
      const eoff_t ee = __shfl(e, leader, 64);
      for (eoff_t k0 = bb; k0 < ee; k0 += 64) {
        const eoff_t k = k0 + lane;
        vis.edge(leader, k, k < ee);
      }
    }
    if (deg >= 64u) deg = 0;
  }

  // ---- tier 3: small rows, packed
  {
    const unsigned incl = gdn_wave_incl_scan(deg);
    const unsigned total = __shfl(incl, 63, 64);
    if (total) {
      s_scan[lane] = incl - deg;  // exclusive prefix
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (unsigned base = 0; base < total; base += 64) {
        const unsigned idx = base + lane;
        const bool valid = idx < total;
        // largest owner with excl[owner] <= idx (rows of degree 0 share a prefix: skip them by
        // taking the LAST lane whose prefix is <= idx)
        int lo = 0, hi = 63;
#pragma unroll
        for (int s = 0; s < 6; s++) {
          const int mid = (lo + hi + 1) >> 1;
          if (s_scan[mid] <= idx) lo = mid;
          else hi = mid - 1;
        }
        const int owner = valid ? lo : (int)lane;
        const eoff_t ob = __shfl(b, owner, 64);
        const unsigned oex = s_scan[owner];
        vis.edge(owner, ob + (idx - oex), valid);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// Persistent consumer of the big-row work items: one wave per item.
template <class V>
__device__ __forceinline__ void gdn_expand_big_items(const eoff_t *__restrict__ rowptr, ExpBigList big,
                                                     V &vis) {
  const unsigned lane = gdn_lane();
  const unsigned nwaves = gridDim.x * GDN_WAVES_PER_BLOCK;
  const unsigned gw = blockIdx.x * GDN_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  unsigned n = *big.count;
  if (n > big.capacity) n = big.capacity;
  for (unsigned it = gw; it < n; it += nwaves) {
    const unsigned long long item = big.items[it];
    const vid_t v = (vid_t)(unsigned)(item & 0xFFFFFFFFull);
    const unsigned c = (unsigned)(item >> 32);
    const eoff_t rb = rowptr[v], re = rowptr[v + 1];
    const eoff_t bb = rb + (eoff_t)c * EXP_CHUNK;
    const eoff_t ee = (bb + EXP_CHUNK < re) ? bb + EXP_CHUNK : re;
    vis.begin_big(v);
    for (eoff_t k0 = bb; k0 < ee; k0 += 64) {
      const eoff_t k = k0 + lane;
      vis.edge(0, k, k < ee);
    }
  }
}
#endif  // __HIPCC__
