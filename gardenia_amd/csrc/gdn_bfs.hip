// gdn_bfs.hip -- breadth-first search: data-driven top-down + bitmap bottom-up, switched by
// Beamer's alpha/beta rule.
//
// Reference path: BFSSolver (src/bfs/bfs.h:43).  Control flow and constants follow the
// direction-optimising OpenMP solver src/bfs/omp_beamer.cc:97-160 (alpha=15, beta=18 :111;
// TDStep :35-56; BUStep :13-31; QueueToBitmap :58 / BitmapToQueue :66); when no reverse graph
// is given it degenerates to the level loop of src/bfs/omp_base.cc:51-56.  The CUDA twins it
// supersedes: src/bfs/linear_base.cu:9 (thread per frontier vertex, one global atomicAdd per
// discovered vertex), linear_lb.cu:130 (CTA/warp/scan expand), hybrid_base.cu:12-58.
//
// MI355X specifics: the visited set is a bitmap (2^27 vertices = 16 MiB, mostly L2/MALL
// resident) probed before the 4-byte depth array is touched; discovery = device-scope
// atomicOr on the bitmap word (coherent across the 8 XCD L2s), so a stale plain probe can
// only cause a redundant atomic, never a wrong depth; next-frontier compaction is one
// atomicAdd per wavefront (gdn_wl_push).  Depths are exact: every vertex is claimed once, in
// the level in which it is first reached.
// The resident plan (gdn_bfs_plan_*) adds, from the sizes on at which each was measured to pay: head records of the bottom-up
// step (every row's in-neighbour of highest out-degree, hubs named by rank; a compact copy for the rows with in-edges),
// a dense sweep and a binned top-down level for heavy frontiers, fused light levels, and DEFERRED depths -- heavy levels keep
// their frontier bitmaps and one sequential pass writes the distances at the end (bfs_depth_finish_kernel).  DESIGN.md 4.4.
#include <stdlib.h>
#include <string.h>

// the top-down kernels' staging strips: 1024 entries per wave (16 KB per workgroup) instead of the library's 256 -- a flush is one
// reservation on the queue's tail, and a line serves 88 M of those per second (tools/atomic_probe.hip): a level that discovers
// 4.5 M vertices was 0.437 ms with 256, 0.396 with 512, 0.362 with 1024 (RMAT-27, profiles/sessions/r06_27.sh)
#define GDN_WL_STAGE 1024
#include "gdn_expand.hpp"
#include "gdn_pb.hpp"

// Every hot counter on a 128-byte line of its own: atomics on one line serialise at ~12-25 ns each whatever issues them,
// lines of their own go through different L2 channels side by side (gdn_sssp.hip: a 228 K-vertex pass 0.26 -> 0.11 ms).
struct BfsCounters {  // device, zeroed per level by the host-side memset
  alignas(128) unsigned next_count;
  alignas(128) unsigned big_count;
  alignas(128) unsigned long long scout;  // sum of out-degrees of the vertices discovered this level
  alignas(128) unsigned long long awake;  // vertices discovered by a bottom-up step
  alignas(128) unsigned overflow;
  alignas(128) unsigned long long bu_by_head;  // bottom-up: rows discovered through their hub head ...
  unsigned long long bu_probes;                //            ... and in-neighbours probed for the others (GDN_BFS_TRACE)
};

#define BFS_REC_LONE (1ull << 63)                                      // head record: the head is the only in-neighbour
#define BFS_REC_DEG(rc) ((unsigned long long)(((rc) >> 32) & 0x7FFFFFFFull))  // head record: the row's out-degree
struct BfsTdVis {
  const eoff_t *__restrict__ rowptr;
  const unsigned long long *__restrict__ rec;  // nullable: the plan's head records (bfs_hub_head_kernel): a discovered vertex's
                                               // out-degree is ONE 8-byte gather there, two through the row offsets
  const vid_t *__restrict__ colidx;
  unsigned *__restrict__ visited;
  int32_t *__restrict__ depth;
  vid_t *__restrict__ outq;
  BfsCounters *cnt;
  unsigned cap;
  int32_t next_level;
  unsigned long long scout_local;
  bool blind = false;  // no read of the bitmap in front of the atomic (see edge())
  GdnWlStage stage;  // per-wave LDS strip of discovered vertices (one atomicAdd on next_count per flush)
  __device__ __forceinline__ void begin_big(vid_t) {}
  __device__ __forceinline__ void edge(int, eoff_t k, bool valid) {
    bool claim = false;
    vid_t dst = 0;
    if (valid) {
      dst = __builtin_nontemporal_load(colidx + k);
      const unsigned bit = 1u << (dst & 31);
      // the plain read in front of the atomic keeps the edges of visited vertices (and the thousands of edges into one hub: a
      // line serves 88 M atomics/s) off the atomic unit.  Where hardly anything is visited yet and no vertex is hot -- the early
      // levels of a graph without hubs -- it is a gather for nothing: `blind` (host: !skewed and < 1/8 of the rows visited)
      const unsigned w = blind ? 0u : visited[dst >> 5];
      if (!(w & bit)) {
        const unsigned old = atomicOr(&visited[dst >> 5], bit);
        claim = !(old & bit);
      }
    }
    if (claim) {
      if (depth) depth[dst] = next_level;  // (null: deferred -- the level's bitmap is visited ^ snapshot, bfs_depth_finish_kernel)
      scout_local += rec ? (eoff_t)BFS_REC_DEG(rec[dst]) : rowptr[dst + 1] - rowptr[dst];
    }
    gdn_wl_push_staged(stage, outq, &cnt->next_count, cap, claim, dst, &cnt->overflow);
  }
  // by the whole workgroup at the end of the kernel: one queue reservation and one counter add per WORKGROUP (per wave, the
  // 8192 waves of the big-row kernel put 16 K atomics on two addresses behind a level that discovers 3 M vertices)
  __device__ __forceinline__ void finish(unsigned *s_tmp, unsigned long long *s_tmp64) {
    gdn_wl_flush_block(stage, outq, &cnt->next_count, cap, &cnt->overflow, s_tmp);
    gdn_block_add_u64(scout_local, &cnt->scout, s_tmp64);
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
bfs_td_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ inq, unsigned nf, ExpBigList big,
              BfsTdVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  __shared__ vid_t s_stage[GDN_WAVES_PER_BLOCK][GDN_WL_STAGE];
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vid_t v = 0;
  if (i < nf) {
    v = inq[i];
    b = rowptr[v];
    e = rowptr[v + 1];
  }
  vis.scout_local = 0;
  vis.stage.strip = s_stage[threadIdx.x >> 6];
  vis.stage.n = 0;
  gdn_expand_wave(b, e, v, big, vis, s_scan[threadIdx.x >> 6]);
  __shared__ unsigned s_tmp[GDN_WAVES_PER_BLOCK + 1];
  __shared__ unsigned long long s_tmp64[GDN_WAVES_PER_BLOCK];
  vis.finish(s_tmp, s_tmp64);
}

__global__ void __launch_bounds__(GDN_BLOCK)
bfs_td_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, BfsTdVis vis) {
  __shared__ vid_t s_stage[GDN_WAVES_PER_BLOCK][GDN_WL_STAGE];
  __shared__ unsigned s_tmp[GDN_WAVES_PER_BLOCK + 1];
  __shared__ unsigned long long s_tmp64[GDN_WAVES_PER_BLOCK];
  vis.stage.strip = s_stage[threadIdx.x >> 6];
  vis.stage.n = 0;
  vis.scout_local = 0;
  gdn_expand_big_items(rowptr, big, vis);
  vis.finish(s_tmp, s_tmp64);
}

// ---- fused light levels (the "fusion" variant of the reference, src/bfs/fusion.cu, without its grid barrier): ONE
// workgroup runs consecutive top-down levels for as long as the frontier stays tiny -- a wave per frontier vertex, the
// queues ping-pong in global memory, the level boundary is a __syncthreads() -- and returns to the host when the frontier
// outgrows one CU (or empties).  A level costs a few dependent memory round trips here (~5 us) instead of a memset, two
// launches and a blocking read back (~55 us): the first and last levels of every search, and every level of a
// high-diameter graph (a 150 000-vertex chain took 9 s level by level).
#define BFS_SMALL_THREADS 1024
struct BfsSmallOut {
  unsigned levels;      // levels expanded
  unsigned nf;          // size of the frontier left for the host (0 = the search is over)
  unsigned which;       // queue that holds it (0 = q0, 1 = q1)
  unsigned overflow;
  unsigned long long scout;       // sum of out-degrees of that frontier
  unsigned long long checked;     // sum of the scout counts of the frontiers expanded here
  unsigned long long discovered;  // vertices discovered here
};

__global__ void __launch_bounds__(BFS_SMALL_THREADS)
bfs_td_small_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, unsigned *__restrict__ visited,
                    int32_t *__restrict__ depth, vid_t *q0, vid_t *q1, unsigned which, unsigned nf, unsigned cap,
                    int32_t level, unsigned long long scout_cur, unsigned max_nf, unsigned long long max_scout,
                    unsigned max_levels, BfsSmallOut *__restrict__ out) {
  __shared__ unsigned s_cnt, s_over;
  __shared__ unsigned long long s_scout;
  const unsigned lane = gdn_lane(), wave = threadIdx.x >> 6, nwaves = BFS_SMALL_THREADS / 64;
  unsigned levels = 0;
  unsigned long long checked = 0, discovered = 0;
  if (threadIdx.x == 0) s_over = 0u;
  for (;;) {
    if (threadIdx.x == 0) {
      s_cnt = 0u;
      s_scout = 0ull;
    }
    __syncthreads();
    const vid_t *qin = which ? q1 : q0;
    vid_t *qout = which ? q0 : q1;
    unsigned long long scout = 0;
    for (unsigned i = wave; i < nf; i += nwaves) {
      // the queues are written and read by this workgroup only, but through different waves: device-scope loads
      const vid_t v = __hip_atomic_load(qin + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const eoff_t b = rowptr[v], e = rowptr[v + 1];
      for (eoff_t k0 = b; k0 < e; k0 += 64) {
        const eoff_t k = k0 + lane;
        bool claim = false;
        vid_t dst = 0;
        if (k < e) {
          dst = colidx[k];
          const unsigned bit = 1u << (dst & 31);
          if (!(visited[dst >> 5] & bit)) claim = !(atomicOr(&visited[dst >> 5], bit) & bit);
        }
        const unsigned long long mask = __ballot(claim);
        if (mask) {
          unsigned base = 0;
          if (lane == 0) base = atomicAdd(&s_cnt, (unsigned)__popcll(mask));
          base = __shfl(base, 0, 64);
          if (claim) {
            depth[dst] = level + 1;
            scout += rowptr[dst + 1] - rowptr[dst];
            const unsigned pos = base + (unsigned)__popcll(mask & gdn_lanemask_lt());
            if (pos < cap) __hip_atomic_store(qout + pos, dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else s_over = 1u;
          }
        }
      }
    }
    scout = gdn_wave_sum(scout);
    if (lane == 0 && scout) atomicAdd(&s_scout, scout);
    gdn_wg_level_sync();
    checked += scout_cur;
    nf = s_cnt;
    scout_cur = s_scout;
    discovered += nf;
    which ^= 1u;
    level++;
    levels++;
    __syncthreads();  // everybody has read s_cnt / s_scout before the next level resets them
    if (nf == 0 || nf > max_nf || scout_cur > max_scout || levels >= max_levels || s_over) break;
  }
  if (threadIdx.x == 0) {
    out->levels = levels;
    out->nf = nf;
    out->which = which;
    out->overflow = s_over;
    out->scout = scout_cur;
    out->checked = checked;
    out->discovered = discovered;
  }
}

// Light levels of a high-diameter graph beyond one workgroup: the same loop on a COOPERATIVE grid -- one workgroup per CU,
// every level ends in a grid barrier (cooperative_groups::grid_group::sync, the CDNA form of the reference's software
// global barrier include/gbar.h:24-65 under its persistent "fusion" kernels, src/bfs/fusion.cu:163-178) instead of a
// kernel boundary and a blocking read back.  A 4096 x 4096 road-like grid (8 191 levels of <= 4 096 vertices) is the
// case: 28 us per level on the host loop.  Three counter sets rotate (the set of level L+1 is zeroed while level L runs;
// the set of level L-1 may still be read by a late workgroup), each counter on a cache line of its own.
#define BFS_COOP_THREADS 256
struct BfsCoopCnt {
  alignas(128) unsigned count;
  alignas(128) unsigned long long scout;
  alignas(128) unsigned over;
};

// (the barrier: gdn_grid_barrier, gdn_common.hpp; cooperative_groups' grid.sync() cost ~30 us per level here)
__global__ void __launch_bounds__(BFS_COOP_THREADS)
bfs_td_coop_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, unsigned *__restrict__ visited,
                   int32_t *__restrict__ depth, vid_t *q0, vid_t *q1, unsigned which, unsigned nf, unsigned cap, int32_t level,
                   unsigned long long scout_cur, unsigned max_nf, unsigned long long max_scout, unsigned min_nf,
                   BfsCoopCnt *cnt /* 3 sets, zeroed by the host */, unsigned *bar /* GDN_GBAR_WORDS, zeroed by the host */,
                   BfsSmallOut *__restrict__ out) {
  const unsigned lane = gdn_lane();
  const unsigned gt = blockIdx.x * BFS_COOP_THREADS + threadIdx.x, nt = gridDim.x * BFS_COOP_THREADS;
  unsigned levels = 0;
  unsigned long long checked = 0, discovered = 0;
  bool over = false;
  for (;;) {
    BfsCoopCnt *cur = cnt + (levels % 3u);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      BfsCoopCnt *nxt = cnt + ((levels + 1u) % 3u);
      __hip_atomic_store(&nxt->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&nxt->scout, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const vid_t *qin = which ? q1 : q0;
    vid_t *qout = which ? q0 : q1;
    unsigned long long scout = 0;
    // one edge of `dst` per lane and step: claim it, push the claimed ones with one counter add per wave step
    auto visit = [&](bool valid, vid_t dst) {
      bool claim = false;
      if (valid) {
        const unsigned bit = 1u << (dst & 31);
        if (!(__hip_atomic_load(visited + (dst >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit))
          claim = !(atomicOr(&visited[dst >> 5], bit) & bit);
      }
      const unsigned long long mask = __ballot(claim);
      if (mask) {
        const int leader = __ffsll((long long)mask) - 1;
        unsigned base = 0;
        if ((int)lane == leader) base = atomicAdd(&cur->count, (unsigned)__popcll(mask));
        base = __shfl(base, leader, 64);
        if (claim) {
          depth[dst] = level + 1;
          scout += rowptr[dst + 1] - rowptr[dst];
          const unsigned pos = base + (unsigned)__popcll(mask & gdn_lanemask_lt());
          if (pos < cap) __hip_atomic_store(qout + pos, dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else over = true;
        }
      }
    };
    for (unsigned i0 = gt - lane; i0 < nf; i0 += nt) {  // a wave takes 64 frontier vertices, one per lane
      const unsigned i = i0 + lane;
      eoff_t b = 0, e = 0;
      if (i < nf) {
        const vid_t v = __hip_atomic_load(qin + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b = rowptr[v];
        e = rowptr[v + 1];
      }
      // rows of a wave's width or more: the whole wave walks them one after the other
      unsigned long long longs = __ballot(e - b >= 64u);
      const bool mine_long = e - b >= 64u;
      while (longs) {
        const int leader = __ffsll((long long)longs) - 1;
        longs &= longs - 1ull;
        const eoff_t bb = __shfl(b, leader, 64), ee = __shfl(e, leader, 64);
        for (eoff_t k0 = bb; k0 < ee; k0 += 64) {
          const eoff_t k = k0 + lane;
          visit(k < ee, k < ee ? colidx[k] : 0);
        }
      }
      if (mine_long) b = e;
      // short rows: every lane walks its own row (64 rows in flight per wave instead of one)
      for (eoff_t j = 0; __any(b + j < e); j++) {
        const bool valid = b + j < e;
        visit(valid, valid ? colidx[b + j] : 0);
      }
    }
    scout = gdn_wave_sum(scout);
    if (lane == 0 && scout) atomicAdd(&cur->scout, scout);
    if (__any(over) && lane == 0) __hip_atomic_store(&cur->over, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gdn_grid_barrier(bar, gridDim.x);
    checked += scout_cur;
    nf = __hip_atomic_load(&cur->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    scout_cur = __hip_atomic_load(&cur->scout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned ov = __hip_atomic_load(&cur->over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    discovered += nf;
    which ^= 1u;
    level++;
    levels++;
    // leave when the frontier is empty, too heavy for this grid, or light enough for the one-workgroup kernel again
    if (nf == 0 || nf > max_nf || scout_cur > max_scout || nf < min_nf || ov) {
      over = ov != 0u;
      break;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out->levels = levels;
    out->nf = nf;
    out->which = which;
    out->overflow = over ? 1u : 0u;
    out->scout = scout_cur;
    out->checked = checked;
    out->discovered = discovered;
  }
}

// ---- hub heads of the bottom-up step (see bfs_bu_kernel)
#ifndef BFS_HUBS
#define BFS_HUBS (1u << 17)      // hubs tracked: their frontier bits are 16 KB of LDS per workgroup
#endif
#define BFS_NO_HUB 0xFFFFFFFFu
// OUTER hubs (round 5): the ranks [BFS_HUBS, BFS_HUBS2) are named by rank in the head records as well and tested against a
// frontier bitmap indexed by RANK (256 KB: resident in every XCD's L2) instead of the vertex-indexed one (16 MB at RMAT-27: a
// 64-byte line from beyond L2 per probe).  R-MAT: 66 % of the rows have their head among the 2^17 inner hubs, 89 % among 2^21.
#ifndef BFS_HUBS2
#define BFS_HUBS2 (1u << 21)
#endif
// keys (out-degree << 32 | vertex) of every vertex, for the sort that ranks them
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_hub_keys_kernel(const eoff_t *__restrict__ out_rowptr, int32_t m, unsigned long long *__restrict__ keys) {
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (unsigned)m) {
    const eoff_t d = out_rowptr[v + 1] - out_rowptr[v];
    keys[v] = ((unsigned long long)(d > 0xFFFFFFFFull ? 0xFFFFFFFFull : d) << 32) | v;
  }
}
// rank[vertex] = its position in the descending order of out-degrees (the END of the ascending sort is rank 0; no
// out-edge: BFS_NO_HUB, never in a frontier that matters); hub_id[k] = the vertex of rank k < n_hubs
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_hub_rank_kernel(const unsigned long long *__restrict__ sorted, int32_t m, unsigned n_hubs, vid_t *__restrict__ hub_id,
                    unsigned *__restrict__ rank) {
  const unsigned k = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (k >= (unsigned)m) {
    if (k < n_hubs) hub_id[k] = -1;  // fewer vertices than slots
    return;
  }
  const unsigned long long key = sorted[(size_t)m - 1 - k];
  const unsigned v = (unsigned)(key & 0xFFFFFFFFull);
  const bool any = (key >> 32) != 0ull;
  rank[v] = any ? k : BFS_NO_HUB;
  if (k < n_hubs) hub_id[k] = any ? (vid_t)v : -1;
}
// rec[v] = out-degree of v << 32 | HEAD of v: its in-neighbour of highest out-degree -- as the hub index when that is one
// of the BFS_HUBS largest (tested against LDS bits), else as 2^31 | vertex (tested against the frontier bitmap), BFS_NO_HUB
// without in-edges.  One wave per row, over at most BFS_HEAD_SCAN of its in-edges: ANY in-neighbour is a valid head, the
// best one only raises the hit rate, and the rows beyond (0.01 % of RMAT-27's) are hubs themselves, discovered top-down --
// walked whole, the few rows of millions of in-edges made the plan build 2.8 s longer.  (Measured and dropped: the three best hubs, 16 bits each, tested together -- with
// 2^16 hub slots the heavy level of RMAT-27 1.26 -> 1.60 ms, profiles/r03_bfs_bottom_up.txt.)
#define BFS_HEAD_VERTEX 0x80000000u
#define BFS_HEAD_SCAN 8192
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_hub_head_kernel(const eoff_t *__restrict__ in_rowptr, const vid_t *__restrict__ in_colidx,
                    const eoff_t *__restrict__ out_rowptr, int32_t m, const unsigned *__restrict__ rank,
                    const unsigned long long *__restrict__ sorted, unsigned long long *__restrict__ rec, unsigned n_ranked = BFS_HUBS) {
  const unsigned lane = gdn_lane();
  const size_t nwaves = ((size_t)gridDim.x * GDN_BLOCK) >> 6;
  for (size_t v = ((size_t)blockIdx.x * GDN_BLOCK + threadIdx.x) >> 6; v < (size_t)m; v += nwaves) {
    const eoff_t b = in_rowptr[v];
    eoff_t e = in_rowptr[v + 1];
    if (e - b > BFS_HEAD_SCAN) e = b + BFS_HEAD_SCAN;
    unsigned best = BFS_NO_HUB;
    for (eoff_t k = b + lane; k < e; k += 64) {
      const unsigned h = rank[in_colidx[k]];
      best = h < best ? h : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned t = (unsigned)__shfl_xor((int)best, o, 64);
      best = t < best ? t : best;
    }
    if (lane == 0) {
      unsigned code = best;
      if (best != BFS_NO_HUB && best >= n_ranked) code = BFS_HEAD_VERTEX | (unsigned)(sorted[(size_t)m - 1 - best] & 0x7FFFFFFFull);
      const eoff_t d = out_rowptr[v + 1] - out_rowptr[v];
      // bit 63 (out-degrees stay below 2^31): the head is the row's ONLY in-neighbour -- when it is not in the frontier the
      // row cannot be discovered at this level, and the bottom-up step does not queue it for a scan (two row-offset gathers,
      // a neighbour and a frontier word saved per such row and level)
      const unsigned long long lone = (in_rowptr[v + 1] - in_rowptr[v] == 1) ? BFS_REC_LONE : 0ull;
      rec[v] = lone | ((unsigned long long)d << 32) | code;
    }
  }
}
// this level's frontier bits of the hubs
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_hub_front_kernel(const vid_t *__restrict__ hub_id, const unsigned *__restrict__ front, unsigned *__restrict__ hub_front,
                     unsigned *__restrict__ hub_front2 = nullptr) {
  const unsigned k = blockIdx.x * GDN_BLOCK + threadIdx.x;  // grid covers the ranked hubs exactly (BFS_HUBS, or BFS_HUBS2 with outer hubs)
  const vid_t u = hub_id[k];
  const bool in = u >= 0 && ((front[(unsigned)u >> 5] >> ((unsigned)u & 31u)) & 1u);
  const unsigned long long mask = __ballot(in);
  if (hub_front2 && (gdn_lane() & 31u) == 0) hub_front2[k >> 5] = (unsigned)(mask >> (gdn_lane() & 32u));
  if (k >= BFS_HUBS) return;  // (whole workgroups: BFS_HUBS is a multiple of the block size)
  if ((gdn_lane() & 31u) == 0) hub_front[k >> 5] = (unsigned)(mask >> (gdn_lane() & 32u));
  // (how many hubs the frontier holds -- the bottom-up step uses the inner hubs' bits only when that is worth it -- is counted by
  // the workgroups of the step themselves from the 16 KB they load anyway: a counter here cost a memset per level, round 5)
}

// Bottom-up step (omp_beamer.cc:13-31): a row not yet visited is discovered if one of its in-neighbours is in the frontier.
// Round 3 form -- the first one walked the vertices one per thread and iteration, every iteration a chain of dependent
// reads (visited word -> row offsets -> neighbour -> frontier word), 256 iterations per thread at RMAT-27: latency bound at
// 1.4-1.7 ms for the heavy level, and in the late levels three lanes of a wave scanned while 61 idled (0.3 ms for a level
// that discovers 20 000 rows).  A workgroup now takes windows of BFS_BU_STEP consecutive rows:
//   stage 0, one bitmap word per thread (visited | no-in-edges, fetched one step ahead): the OPEN rows of the window are
//            listed in LDS (a block scan of the pop-counts) -- a late level is nothing but this;
//   stage 1, the listed rows, one per thread and dense: HEADS.  rec[v] (8 bytes, bfs_hub_head_kernel) names the
//            in-neighbour of v with the highest out-degree -- by hub index when it is one of the BFS_HUBS largest (this
//            level's frontier bits of the hubs sit in LDS: hub_front), by vertex id else (one read of the frontier bitmap)
//            -- and carries v's out-degree: a row whose head is in the frontier is discovered from ONE 8-byte read, the
//            frontier's out-degree sum comes with it.  RMAT-27, the heavy level: 43.8 M of 44.1 M discoveries (with hub
//            indices alone 35.3 M; the 12 M others then cost 0.57 ms of row-offset / neighbour / frontier gathers at the
//            ~55 G/s gather wall); the rows still open go into an LDS queue that spans up to BFS_BU_WIN windows;
//   stage 2, when the queue is full or BFS_BU_WIN windows wait: every thread takes queued rows and scans their
//            in-neighbours (dense lanes whatever the level); then the windows' next / visited words are written whole --
//            every word belongs to one workgroup and is written once, no atomics on global memory.
// Measured (profiles/r03_bfs_bottom_up.txt): heavy level 1.72 -> 1.03 ms, the level after 0.85 -> 0.37, a late level
// 0.31 -> 0.17; compile-time knobs (windows 16, grid 1024 / 4096, 2 / 8 rows side by side) within 3 %, 2^15 / 2^16 hubs
// 14 % / 8 % slower on the heavy level.
// Persistent grid: the awake / scout totals are kept in registers and added to the level counters ONCE per workgroup.
#define BFS_BU_WORDS 128                                  // bitmap words per window (threads 0..127 fetch one each)
#define BFS_BU_STEP (BFS_BU_WORDS * 32)                   // rows per window
#ifndef BFS_BU_WIN
#define BFS_BU_WIN 8                                      // windows whose open rows may wait together
#endif
#define BFS_BU_Q 4096                                     // queue entries (u16: window << 12 | row in window)
#ifndef BFS_BU_UNR
#define BFS_BU_UNR 4                                      // listed rows per thread whose loads are issued side by side
#endif
#ifndef BFS_BU_GRID
#define BFS_BU_GRID (256 * 8)
#endif
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_bu_kernel(const eoff_t *__restrict__ in_rowptr, const vid_t *__restrict__ in_colidx,
              const eoff_t *__restrict__ out_rowptr, int32_t m, unsigned m_pad, const unsigned *__restrict__ front,
              unsigned *__restrict__ next, unsigned *__restrict__ visited, int32_t *__restrict__ depth,
              int32_t next_level, BfsCounters *cnt,
              // nullable: bitmap of the rows WITHOUT in-edges (never discoverable): skipped without touching their row
              // offsets -- 61 % of RMAT-27's rows, 16 B of in_rowptr each otherwise
              const unsigned *__restrict__ noin = nullptr,
              // hub heads (nullable, see above)
              const unsigned long long *__restrict__ rec = nullptr, const unsigned *__restrict__ hub_front = nullptr,
              unsigned min_hubs = 0, bool trace = false,
              // the outer hubs' frontier bits by rank (null when the head records name inner hubs only)
              const unsigned *__restrict__ hub_front2 = nullptr) {
  static_assert(BFS_BU_STEP == 4096 && BFS_BU_WIN <= 16 && BFS_BU_Q >= BFS_BU_STEP && BFS_BU_WORDS <= GDN_BLOCK,
                "queue entries keep the row in 12 bits, the window in 4; one window's open rows must fit the empty queue");
  __shared__ unsigned long long s_red[4 * GDN_WAVES_PER_BLOCK];
  __shared__ unsigned s_hf[BFS_HUBS / 32];
  __shared__ unsigned short s_list[BFS_BU_STEP];       // open rows of the current window, ascending
  __shared__ unsigned short s_q[BFS_BU_Q];             // rows waiting for their in-neighbour scan
  __shared__ unsigned s_bits[BFS_BU_WIN][BFS_BU_WORDS];  // discoveries of the windows that wait, 1 bit per row
  __shared__ unsigned s_wbase[BFS_BU_WIN];              // first row of every waiting window
  __shared__ unsigned s_wsum[GDN_WAVES_PER_BLOCK];
  __shared__ unsigned s_qn;
  const unsigned lane = gdn_lane(), wave = threadIdx.x >> 6;
  unsigned n_hub_bits = 0;
  if (rec)
    for (unsigned i = threadIdx.x; i < BFS_HUBS / 32; i += GDN_BLOCK) {
      const unsigned wv = hub_front[i];
      s_hf[i] = wv;
      n_hub_bits += (unsigned)__popc(wv);
    }
  n_hub_bits = gdn_wave_sum(n_hub_bits);
  if (lane == 0) s_wsum[wave] = n_hub_bits;
  if (threadIdx.x == 0) s_qn = 0u;
  __syncthreads();
  n_hub_bits = 0;
  for (int i = 0; i < GDN_WAVES_PER_BLOCK; i++) n_hub_bits += s_wsum[i];
  const bool hubs = rec && n_hub_bits >= min_hubs;  // else: too few hubs in this frontier (uniform)
  __syncthreads();  // (s_wsum is used again below)
  unsigned long long awake = 0, scout = 0;
  unsigned by_head = 0, probes = 0;
  unsigned nwin = 0;  // windows waiting (block-uniform)
  // stage 2a: the queued rows, one per thread
  auto scan_queue = [&]() {
    __syncthreads();
    const unsigned n = s_qn;
    // (four queued rows side by side per thread -- offsets, first neighbours, frontier words as rounds of independent
    // loads -- measured no faster: the scans run at the gather wall, not at a latency chain)
    for (unsigned i = threadIdx.x; i < n; i += GDN_BLOCK) {
      const unsigned e = s_q[i], wi = e >> 12, rl = e & 4095u;
      const unsigned v = s_wbase[wi] + rl;
      const eoff_t rb = in_rowptr[v], re = in_rowptr[v + 1];
      bool found = false;
      for (eoff_t k = rb; k < re; k++) {
        const vid_t u = in_colidx[k];
        probes++;
        if ((front[(unsigned)u >> 5] >> ((unsigned)u & 31u)) & 1u) {
          found = true;
          break;
        }
      }
      if (found) {
        depth[v] = next_level;
        atomicOr(&s_bits[wi][rl >> 5], 1u << (rl & 31u));
        awake++;
        scout += rec ? (eoff_t)BFS_REC_DEG(rec[v]) : out_rowptr[v + 1] - out_rowptr[v];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_qn = 0u;
    __syncthreads();
  };
  // stage 2b: the waiting windows' words go out
  auto flush = [&]() {
    scan_queue();
    for (unsigned i = threadIdx.x; i < nwin * BFS_BU_WORDS; i += GDN_BLOCK) {
      const unsigned wi = i / BFS_BU_WORDS, j = i % BFS_BU_WORDS;
      const unsigned word = (s_wbase[wi] >> 5) + j;
      if ((word << 5) < m_pad) {
        const unsigned bits = s_bits[wi][j];
        next[word] = bits;
        if (bits) visited[word] |= bits;
      }
    }
    nwin = 0;
    __syncthreads();
  };
  // the bitmap word of a thread, fetched one window ahead (another workgroup's words never change under this one)
  auto fetch = [&](unsigned base) -> unsigned {
    const unsigned x = base + threadIdx.x * 32u;  // first row of the word
    unsigned w = ~0u;
    if (threadIdx.x < BFS_BU_WORDS && base < m_pad && x < (unsigned)m) {
      w = visited[x >> 5] | (noin ? noin[x >> 5] : 0u);
      if ((unsigned)m - x < 32u) w |= ~0u << ((unsigned)m - x);  // rows past the last one
    }
    return w;
  };
  unsigned w_next = fetch(blockIdx.x * BFS_BU_STEP);
  for (unsigned base = blockIdx.x * BFS_BU_STEP; base < m_pad; base += gridDim.x * BFS_BU_STEP) {  // block-uniform trip count
    // ---- stage 0
    unsigned open = ~w_next;
    w_next = fetch(base + gridDim.x * BFS_BU_STEP);
    const unsigned cnt_open = (unsigned)__popc(open);
    const unsigned incl = gdn_wave_incl_scan(cnt_open);
    if (lane == 63) s_wsum[wave] = incl;
    if (threadIdx.x < BFS_BU_WORDS) s_bits[nwin][threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_wbase[nwin] = base;
    __syncthreads();
    unsigned off = incl - cnt_open, n = 0;
#pragma unroll
    for (unsigned i = 0; i < GDN_WAVES_PER_BLOCK; i++) {
      const unsigned t = s_wsum[i];
      if (i < wave) off += t;
      n += t;
    }
    if (n == 0) {  // nothing open in the window (uniform): its words are zero, and they still have to go out
      nwin++;
      if (nwin == BFS_BU_WIN) flush();
      else __syncthreads();
      continue;
    }
    while (open) {
      const unsigned b = (unsigned)__ffs((int)open) - 1u;
      open &= open - 1u;
      s_list[off++] = (unsigned short)(threadIdx.x * 32u + b);
    }
    if (s_qn + n > BFS_BU_Q) scan_queue();  // uniform; leaves room for every row of this window
    else __syncthreads();
    // ---- stage 1
    for (unsigned i0 = 0; i0 < n; i0 += BFS_BU_UNR * GDN_BLOCK) {
      unsigned rl[BFS_BU_UNR], code[BFS_BU_UNR], fw[BFS_BU_UNR];
      unsigned long long rc[BFS_BU_UNR];
      bool on[BFS_BU_UNR], found[BFS_BU_UNR];
#pragma unroll
      for (int r = 0; r < BFS_BU_UNR; r++) {
        const unsigned i = i0 + (unsigned)r * GDN_BLOCK + threadIdx.x;
        on[r] = i < n;
        rl[r] = on[r] ? s_list[i] : 0u;
        rc[r] = BFS_NO_HUB;
        if (on[r] && rec) rc[r] = rec[base + rl[r]];
        code[r] = (unsigned)rc[r];
      }
#pragma unroll
      for (int r = 0; r < BFS_BU_UNR; r++) {  // heads outside the hub set: their frontier word
        fw[r] = 0u;
        if (code[r] != BFS_NO_HUB && (code[r] & BFS_HEAD_VERTEX)) fw[r] = front[(code[r] & ~BFS_HEAD_VERTEX) >> 5];
        else if (code[r] != BFS_NO_HUB && code[r] >= BFS_HUBS) fw[r] = hub_front2[code[r] >> 5];  // an outer hub: its bit by rank
      }
#pragma unroll
      for (int r = 0; r < BFS_BU_UNR; r++) {
        found[r] = code[r] < BFS_HUBS ? (hubs && ((s_hf[code[r] >> 5] >> (code[r] & 31u)) & 1u)) : (bool)((fw[r] >> (code[r] & 31u)) & 1u);
        if (found[r]) {
          depth[base + rl[r]] = next_level;
          scout += BFS_REC_DEG(rc[r]);
          atomicOr(&s_bits[nwin][rl[r] >> 5], 1u << (rl[r] & 31u));
          by_head++;
        }
      }
#pragma unroll
      for (int r = 0; r < BFS_BU_UNR; r++) {
        // into the queue (one LDS reservation per wave) -- unless the head is the row's only in-neighbour and the test above was
        // the real one (a hub head while this level reads no hub bits has not been tested: the scan does it; a row WITHOUT a
        // head -- its only in-neighbour has no out-edge in the out-CSR, i.e. in_csr is not the exact transpose -- is scanned)
        const bool wait = on[r] && !found[r] && !((rc[r] & BFS_REC_LONE) && code[r] != BFS_NO_HUB && (hubs || code[r] >= BFS_HUBS));
        const unsigned long long om = __ballot(wait);
        if (om) {
          const int first = __ffsll((long long)om) - 1;
          unsigned pos = 0;
          if (lane == (unsigned)first) pos = atomicAdd(&s_qn, (unsigned)__popcll(om));
          pos = __shfl(pos, first, 64);
          if (wait) s_q[pos + (unsigned)__popcll(om & gdn_lanemask_lt())] = (unsigned short)((nwin << 12) | rl[r]);
        }
      }
    }
    nwin++;
    if (nwin == BFS_BU_WIN) flush();  // uniform
    else __syncthreads();
  }
  if (nwin) flush();
  awake += by_head;
  awake = gdn_wave_sum(awake);
  scout = gdn_wave_sum(scout);
  // (the two trace counters: per workgroup like the others -- added per WAVE, 16 K atomics on one line cost a level 0.25 ms)
  const unsigned long long bh = trace ? gdn_wave_sum((unsigned long long)by_head) : 0ull;
  const unsigned long long pr = trace ? gdn_wave_sum((unsigned long long)probes) : 0ull;
  const unsigned w = threadIdx.x >> 6;
  if (lane == 0) {
    s_red[w] = awake;
    s_red[GDN_WAVES_PER_BLOCK + w] = scout;
    s_red[2 * GDN_WAVES_PER_BLOCK + w] = bh;
    s_red[3 * GDN_WAVES_PER_BLOCK + w] = pr;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long a = 0, sc = 0, b = 0, c = 0;
    for (int i = 0; i < GDN_WAVES_PER_BLOCK; i++) {
      a += s_red[i];
      sc += s_red[GDN_WAVES_PER_BLOCK + i];
      b += s_red[2 * GDN_WAVES_PER_BLOCK + i];
      c += s_red[3 * GDN_WAVES_PER_BLOCK + i];
    }
    if (a) {
      atomicAdd(&cnt->awake, a);
      atomicAdd(&cnt->scout, sc);
    }
    if (b | c) {
      atomicAdd(&cnt->bu_by_head, b);
      atomicAdd(&cnt->bu_probes, c);
    }
  }
}

// ------------------------------------------------------------------------------------------
// The same step with the WAVE as the unit (round 4).  In bfs_bu_kernel a window is a workgroup's: four barriers and two or
// three dependent round trips per 4 096 rows, sixteen windows one after the other per workgroup -- ~10 us per window
// whatever is in it (a late level that discovers 27 K rows: 0.17 ms; measured with the two passes split into kernels: the
// head pass alone 0.17 / 0.32 / 0.52 ms for the three levels of an RMAT-27 search, profiles/sessions/r04_48.sh).  Here a wave
// owns 2 048 consecutive rows = 64 bitmap words, one per lane, and shares nothing with the other waves of its workgroup but
// the hubs' frontier bits: no barrier after the prologue.  Per step: the words (fetched one step ahead); nothing open ->
// 64 zero words out, next step (a late level is little else: one round trip per 2 048 rows); else the open rows into the
// wave's LDS list (stage 0), heads (stage 1: every lane busy, the rows that stay open compacted to the front of the list),
// the in-neighbour scan of those (stage 2), then the 64 next / visited words in one store each.
// GDN_BFS_BU_FORM=window keeps bfs_bu_kernel.
// ------------------------------------------------------------------------------------------
// ---- frontier filter of the bottom-up scan (round 5).  A row whose head is not in the frontier walks its in-neighbours and
// tests each against the frontier bitmap: 2^27 bits = 16 MB, served beyond L2 at the L2-miss request rate (55 G/s; RMAT-27, the
// level whose frontier is 70 K hubs: 70 M probes = 1.3 of its 1.87 ms).  While the frontier is SMALL (<= 2^20 vertices) a hashed
// filter of 2^23 bits (1 MB: stays in every XCD's L2, 195 G probes/s) answers "not in the frontier" for all but
// |frontier| / 2^23 of the probes; only those go on to the bitmap.
#define BFS_FILT_LOG 23
__device__ __forceinline__ unsigned bfs_filt_hash(unsigned u) { return (u * 2654435761u) >> (32 - BFS_FILT_LOG); }
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_filt_build_kernel(const unsigned *__restrict__ front, unsigned nwords, unsigned *__restrict__ filt) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i >= nwords) return;
  unsigned w = front[i];
  while (w) {
    const unsigned b = (unsigned)__ffs((int)w) - 1u;
    w &= w - 1u;
    const unsigned h = bfs_filt_hash((i << 5) | b);
    atomicOr(&filt[h >> 5], 1u << (h & 31u));
  }
}

#define BFS_BW_STEP 2048u  // rows per wave step (64 bitmap words, one per lane) = entries of the wave's list
// Round 5: 512-thread workgroups held to 80 registers (BFS_BW_MINW 6): eight waves share one copy of the hubs' frontier bits
// (16 KB), a workgroup takes 51 KB of LDS, three fit a CU = 24 waves where 256-thread workgroups at 97 registers gave 16.  The
// step is a chain of dependent gathers per wave, so waves in flight are what it is made of: RMAT-27's hub-frontier level (source 5)
// 2.66 -> 2.41 ms, the other searches and RMAT-24 -0 .. -2 % (sessions/r05_15.sh; a grid of 6 workgroups per CU: +3 %).
#ifndef BFS_BW_THREADS
#define BFS_BW_THREADS 512
#endif
#define BFS_BW_WAVES (BFS_BW_THREADS / 64)
#ifndef BFS_BW_GRID
#define BFS_BW_GRID (256 * 3)  // what is resident at once (51 KB of LDS per workgroup); round 4, 256 threads: 2048 / 4096 workgroups 3 % slower (sessions/r04_60.sh)
#endif
#ifndef BFS_BW_GROUP
#define BFS_BW_GROUP 1u    // steps a wave takes together (their open rows share one pass through the stages when they fit the list).
                           // Measured on RMAT-27: 8 -> a late level 0.13 -> 0.10 ms, but the heavy one 0.84 -> 0.92 (sessions/r04_50.sh): 1
#endif
#ifndef BFS_BW_MINW
#define BFS_BW_MINW 6  // minimum waves per SIMD the register allocation is held to (3 workgroups of 512 threads per CU)
#endif
__global__ void __launch_bounds__(BFS_BW_THREADS, BFS_BW_MINW)
bfs_bu_wave_kernel(const eoff_t *__restrict__ in_rowptr, const vid_t *__restrict__ in_colidx, int32_t m, unsigned m_pad,
                   const unsigned *__restrict__ front, unsigned *__restrict__ next, unsigned *__restrict__ visited,
                   int32_t *__restrict__ depth, int32_t next_level, BfsCounters *cnt, const unsigned *__restrict__ noin,
                   const unsigned long long *__restrict__ rec, const unsigned *__restrict__ hub_front, unsigned min_hubs, bool trace,
                   int scan_unr = 1, const unsigned *__restrict__ filt = nullptr, const unsigned *__restrict__ hub_front2 = nullptr,
                   // non-null: `rec` is the COMPACT copy (rows with in-edges only; see bfs_compact_rec_kernel), cbase its word offsets
                   const unsigned *__restrict__ cbase = nullptr, const eoff_t *__restrict__ rowptr_c = nullptr) {
  static_assert(BFS_BW_GROUP * BFS_BW_STEP <= 65536u, "a list entry is (step << 11 | row in step) in 16 bits");
  auto compact_id = [&](unsigned v) -> unsigned {  // (v is an open row: it has in-edges)
    const unsigned wd = v >> 5;
    return cbase[wd] + (unsigned)__popc(~noin[wd] & ((1u << (v & 31u)) - 1u));
  };
  auto rec_at = [&](unsigned v) -> unsigned long long { return cbase == nullptr ? rec[v] : rec[compact_id(v)]; };
  __shared__ unsigned s_hf[BFS_HUBS / 32];
  __shared__ unsigned short s_list[BFS_BW_WAVES][BFS_BW_STEP];
  __shared__ unsigned s_bits[BFS_BW_WAVES][BFS_BW_GROUP * 64];
  __shared__ unsigned long long s_red[4 * BFS_BW_WAVES];
  const unsigned lane = gdn_lane(), wave = threadIdx.x >> 6;
  unsigned n_hub_bits = 0;
  for (unsigned i = threadIdx.x; i < BFS_HUBS / 32; i += BFS_BW_THREADS) {
    const unsigned wv = hub_front[i];
    s_hf[i] = wv;
    n_hub_bits += (unsigned)__popc(wv);
  }
  __shared__ unsigned s_hub_n[BFS_BW_WAVES];
  n_hub_bits = gdn_wave_sum(n_hub_bits);
  if (lane == 0) s_hub_n[wave] = n_hub_bits;
  __syncthreads();  // the only barrier
  n_hub_bits = 0;
  for (unsigned i = 0; i < BFS_BW_WAVES; i++) n_hub_bits += s_hub_n[i];
  const bool hubs = n_hub_bits >= min_hubs;  // else: too few hubs in this frontier (uniform)
  unsigned short *list = s_list[wave];
  unsigned *bits = s_bits[wave];
  const unsigned nsteps = m_pad / BFS_BW_STEP;  // (m_pad is a multiple of 2 048: nwords_pad of 64)
  const unsigned ngroups = (nsteps + BFS_BW_GROUP - 1) / BFS_BW_GROUP;
  const unsigned gw = blockIdx.x * BFS_BW_WAVES + wave, nw = gridDim.x * BFS_BW_WAVES;
  unsigned long long awake = 0, scout = 0;
  unsigned by_head = 0, probes = 0;
  auto wave_sync = [&]() {  // LDS writes of the wave visible to the wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // the n rows listed in `list` (entry = step in the group << 11 | row in the step) through stages 1 and 2; their discoveries
  // go into `bits` (word = entry >> 5)
  auto run_list = [&](unsigned gbase, unsigned n) {
    // ---- stage 1: heads; the rows that stay open move to the front of the list (written behind what has been read)
    unsigned nq = 0;  // uniform
    for (unsigned i0 = 0; i0 < n; i0 += BFS_BU_UNR * 64u) {
      unsigned rl[BFS_BU_UNR], code[BFS_BU_UNR], fw[BFS_BU_UNR];
      unsigned long long rc[BFS_BU_UNR];
      bool on[BFS_BU_UNR], found[BFS_BU_UNR];
#pragma unroll
      for (int r = 0; r < BFS_BU_UNR; r++) {
        const unsigned i = i0 + (unsigned)r * 64u + lane;
        on[r] = i < n;
        rl[r] = on[r] ? list[i] : 0u;
        rc[r] = BFS_NO_HUB;
        if (on[r]) rc[r] = rec_at(gbase + rl[r]);
        code[r] = (unsigned)rc[r];
      }
#pragma unroll
      for (int r = 0; r < BFS_BU_UNR; r++) {  // heads outside the hub set: their frontier word
        fw[r] = 0u;
        if (code[r] != BFS_NO_HUB && (code[r] & BFS_HEAD_VERTEX)) fw[r] = front[(code[r] & ~BFS_HEAD_VERTEX) >> 5];
        else if (code[r] != BFS_NO_HUB && code[r] >= BFS_HUBS) fw[r] = hub_front2[code[r] >> 5];  // an outer hub: its bit by rank
      }
      wave_sync();  // every lane holds its entries: the front of the list may be overwritten
#pragma unroll
      for (int r = 0; r < BFS_BU_UNR; r++) {
        found[r] = on[r] && (code[r] < BFS_HUBS ? (hubs && ((s_hf[code[r] >> 5] >> (code[r] & 31u)) & 1u)) : (bool)((fw[r] >> (code[r] & 31u)) & 1u));
        if (found[r]) {
          if (depth) depth[gbase + rl[r]] = next_level;  // (null: this level's depths are written from its bitmap at the end of the search)
          scout += BFS_REC_DEG(rc[r]);
          atomicOr(&bits[rl[r] >> 5], 1u << (rl[r] & 31u));
          by_head++;
        }
        // (a lone head outside the frontier: nothing left to scan -- if the head WAS tested: a hub head needs this level's hub bits)
        const bool wait = on[r] && !found[r] && !((rc[r] & BFS_REC_LONE) && code[r] != BFS_NO_HUB && (hubs || code[r] >= BFS_HUBS));
        const unsigned long long om = __ballot(wait);
        if (wait) list[nq + (unsigned)__popcll(om & gdn_lanemask_lt())] = (unsigned short)rl[r];
        nq += (unsigned)__popcll(om);
      }
    }
    wave_sync();
#if defined(BFS_ABL) && BFS_ABL == 1  // timing-only ablation: no stage 2 at all
    nq = 0;
#endif
    // ---- stage 2: the rows still open: their in-neighbours against the frontier
    // FLAT form (scan_unr == 0, the default since round 6): the in-edges of 64 open rows as ONE list, 64 of them per wave step,
    // whichever rows they belong to (owner of a position: binary search over the lanes' running degree sums, six shuffles) --
    // every step is one round trip of independent gathers.  With a row per lane each lane walked its own list, one dependent
    // load per in-neighbour, and a wave step lasted as long as its LONGEST row: on the 0.13-share level of RMAT-27's slow source
    // (38 M open rows of ~3 in-neighbours, 35 M of them failing) that loop alone was 0.95 of the level's 1.34 ms, the probes
    // 0.03 (compile-time ablations, profiles/r06_bfs_slow_source.md).  A row found in one step is skipped in the next.
    if (scan_unr == 0) {
      constexpr unsigned CAP = 1u << 24;  // in-edges of a row the flat pass takes (the sum over 64 rows stays below 2^32); the rest: below
      // (four steps of gathers in flight and the next 64 rows' offsets requested ahead were measured: the same on RMAT-27, 5 % slower
      // on RMAT-22 -- sessions r06_35 / r06_36; the step is no chain of round trips any more)
      for (unsigned i0 = 0; i0 < nq; i0 += 64u) {  // uniform
        const unsigned i = i0 + lane;
        const bool on = i < nq;
        const unsigned rl = on ? list[i] : 0u, v = gbase + rl;
        eoff_t rb = 0, re = 0;
        if (on) {
          if (cbase == nullptr) {
            rb = in_rowptr[v];
            re = in_rowptr[v + 1];
          } else {
            const unsigned ci = compact_id(v);
            rb = rowptr_c[ci];
            re = rowptr_c[ci + 1u];
          }
        }
#if defined(BFS_ABL) && BFS_ABL == 2
        re = rb;
#endif
        const unsigned d = re - rb < (eoff_t)CAP ? (unsigned)(re - rb) : CAP;
        // in ROUNDS: every open row puts its next `quota` in-edges into the round's list, the quota doubles from round to round
        // (4, 8, 16, ...).  A row that finds a parent stops there -- with all in-edges of a row in one list a row of 16
        // in-neighbours paid 16 probes where the frontier owns a quarter of the edges and 4 find a parent (a uniform random
        // graph's heavy level as a bottom-up step: 6.2 ms against the lane-private loop's 2.9, session r06_44) -- and a long
        // failing row is through in log2 rounds; the failing rows of ~3 in-neighbours of an R-MAT level take one round as before.
        unsigned pos = 0, quota = 4u;
        bool found = false;
        for (;;) {
          const unsigned left = d - pos;
          const unsigned q = (!found && left) ? (left < quota ? left : quota) : 0u;
          if (__ballot(q != 0u) == 0ull) break;
          const unsigned incl = gdn_wave_incl_scan(q), excl = incl - q;
          const unsigned total = (unsigned)__shfl((int)incl, 63, 64);
          unsigned long long fmask = 0ull;  // lanes whose row has been found in this round (uniform)
          for (unsigned t0 = 0; t0 < total; t0 += 64u) {
            const unsigned idx = t0 + lane;
            const bool valid = idx < total;
            unsigned owner = 0;  // lanes whose running sum is <= idx = the first lane whose sum exceeds it
#pragma unroll
            for (unsigned sft = 32u; sft > 0u; sft >>= 1) {
              const unsigned at = (unsigned)__shfl((int)incl, (int)((owner + sft - 1u) & 63u), 64);
              if (at <= idx && owner + sft <= 63u) owner += sft;
            }
            const eoff_t ob = __shfl(rb, (int)owner, 64);
            const unsigned oe = (unsigned)__shfl((int)excl, (int)owner, 64) - (unsigned)__shfl((int)pos, (int)owner, 64);
            bool hit = false;
            if (valid && !((fmask >> owner) & 1ull)) {
              const vid_t u = in_colidx[ob + (eoff_t)(idx - oe)];
              probes++;
#if defined(BFS_ABL) && BFS_ABL == 3
              hit = u == (vid_t)-5;
#else
              bool pass = true;
              if (filt) {  // small frontier: its 1 MB hashed filter (L2 resident) first; most in-neighbours of a failing row stop here
                const unsigned h = bfs_filt_hash((unsigned)u);
                pass = (filt[h >> 5] >> (h & 31u)) & 1u;
              }
              if (pass) hit = (front[(unsigned)u >> 5] >> ((unsigned)u & 31u)) & 1u;
#endif
            }
            unsigned long long hm = __ballot(hit);
            while (hm) {  // (the owners of the hits: a few per step)
              const int l = __ffsll((long long)hm) - 1;
              hm &= hm - 1ull;
              fmask |= 1ull << (unsigned)__builtin_amdgcn_readlane((int)owner, l);
            }
          }
          found = found || ((fmask >> lane) & 1ull);
          pos += q;
          quota = quota < (1u << 20) ? quota << 1 : quota;
        }
        found = on && found;
        if (on && !found && re - rb > (eoff_t)CAP) {  // (a row of more than 2^24 in-edges: the rest of its list, one by one)
          for (eoff_t k = rb + CAP; k < re && !found; k++) {
            const vid_t u = in_colidx[k];
            probes++;
            found = (front[(unsigned)u >> 5] >> ((unsigned)u & 31u)) & 1u;
          }
        }
        if (found) {
          if (depth) depth[v] = next_level;
          atomicOr(&bits[rl >> 5], 1u << (rl & 31u));
          awake++;
          scout += BFS_REC_DEG(rec_at(v));
        }
      }
    } else
    for (unsigned i = lane; i < nq; i += 64u) {
      const unsigned rl = list[i], v = gbase + rl;
      eoff_t rb, re;
      if (cbase == nullptr) {
        rb = in_rowptr[v];
        re = in_rowptr[v + 1];
      } else {
        const unsigned ci = compact_id(v);
        rb = rowptr_c[ci];
        re = rowptr_c[ci + 1u];
      }
      bool found = false;
#if defined(BFS_ABL) && BFS_ABL == 2  // timing-only ablation: the open rows' offsets are loaded, nothing is scanned
      found = (re - rb) == 0x7FFFFFFFull;
      re = rb;
#endif
      if (scan_unr > 1) {
        // four in-neighbours and their frontier words per round trip: a row that reaches this stage mostly FAILS (its head was
        // not in the frontier), so the whole list is walked anyway -- one neighbour at a time that is two dependent loads each
        for (eoff_t k = rb; k < re && !found; k += 4) {
          vid_t u[4];
          unsigned fw4[4];
#pragma unroll
          for (int j = 0; j < 4; j++) u[j] = k + j < re ? in_colidx[k + j] : (vid_t)-1;
#pragma unroll
          for (int j = 0; j < 4; j++) fw4[j] = u[j] >= 0 ? front[(unsigned)u[j] >> 5] : 0u;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            probes += u[j] >= 0 ? 1u : 0u;
            found = found || ((fw4[j] >> ((unsigned)u[j] & 31u)) & 1u);
          }
        }
      } else
      for (eoff_t k = rb; k < re; k++) {
        const vid_t u = in_colidx[k];
        probes++;
#if defined(BFS_ABL) && BFS_ABL == 3  // timing-only ablation: the in-neighbours are loaded, nothing is probed
        found = found || u == (vid_t)-5;
        continue;
#endif
        if (filt) {  // small frontier: its 1 MB hashed filter (L2 resident) first; most in-neighbours of a failing row stop here
          const unsigned h = bfs_filt_hash((unsigned)u);
          if (!((filt[h >> 5] >> (h & 31u)) & 1u)) continue;
        }
        if ((front[(unsigned)u >> 5] >> ((unsigned)u & 31u)) & 1u) {
          found = true;
          break;
        }
      }
      if (found) {
        if (depth) depth[v] = next_level;
        atomicOr(&bits[rl >> 5], 1u << (rl & 31u));
        awake++;
        scout += BFS_REC_DEG(rec_at(v));
      }
    }
    wave_sync();
  };
  // the words of a group, fetched one group ahead (another wave's words never change under this one)
  auto fetch = [&](unsigned g, unsigned (&w)[BFS_BW_GROUP]) {
#pragma unroll
    for (unsigned k = 0; k < BFS_BW_GROUP; k++) {
      w[k] = ~0u;
      const unsigned st = g * BFS_BW_GROUP + k;
      if (g < ngroups && st < nsteps) {
        const unsigned idx = st * 64u + lane, x = idx << 5;  // first row of the word
        if (x < (unsigned)m) {
          w[k] = visited[idx] | (noin ? noin[idx] : 0u);
          if ((unsigned)m - x < 32u) w[k] |= ~0u << ((unsigned)m - x);  // rows past the last one
        }
      }
    }
  };
  unsigned w_next[BFS_BW_GROUP];
  fetch(gw, w_next);
  for (unsigned g = gw; g < ngroups; g += nw) {  // wave-uniform trip count
    const unsigned st0 = g * BFS_BW_GROUP, gbase = st0 * BFS_BW_STEP;
    unsigned open[BFS_BW_GROUP];
    unsigned c_tot = 0;
#pragma unroll
    for (unsigned k = 0; k < BFS_BW_GROUP; k++) {
      open[k] = ~w_next[k];
      c_tot += (unsigned)__popc(open[k]);
    }
    fetch(g + nw, w_next);
    const unsigned incl = gdn_wave_incl_scan(c_tot);
    const unsigned n = (unsigned)__shfl((int)incl, 63, 64);
    if (n == 0u) {  // uniform: nothing open in 16 K rows
#pragma unroll
      for (unsigned k = 0; k < BFS_BW_GROUP; k++)
        if (st0 + k < nsteps) next[(st0 + k) * 64u + lane] = 0u;
      continue;
    }
#pragma unroll
    for (unsigned k = 0; k < BFS_BW_GROUP; k++) bits[k * 64u + lane] = 0u;
    if (n <= BFS_BW_STEP) {
      // a light group (any late level): ALL its open rows in one list, one pass through the stages -- the chain of dependent
      // round trips is paid once per 16 K rows
      unsigned off = incl - c_tot;
#pragma unroll
      for (unsigned k = 0; k < BFS_BW_GROUP; k++) {
        unsigned o = open[k];
        while (o) {
          const unsigned b = (unsigned)__ffs((int)o) - 1u;
          o &= o - 1u;
          list[off++] = (unsigned short)((k << 11) | (lane * 32u + b));
        }
      }
      wave_sync();
      run_list(gbase, n);
    } else {
      // a heavy group: step by step (ascending rows per list: the head records of a wave instruction lie together)
#pragma unroll
      for (unsigned k = 0; k < BFS_BW_GROUP; k++) {
        const unsigned ck = (unsigned)__popc(open[k]);
        const unsigned ik = gdn_wave_incl_scan(ck);
        const unsigned nk = (unsigned)__shfl((int)ik, 63, 64);
        if (nk == 0u) continue;  // uniform
        unsigned off = ik - ck, o = open[k];
        while (o) {
          const unsigned b = (unsigned)__ffs((int)o) - 1u;
          o &= o - 1u;
          list[off++] = (unsigned short)((k << 11) | (lane * 32u + b));
        }
        wave_sync();
        run_list(gbase, nk);
      }
    }
#pragma unroll
    for (unsigned k = 0; k < BFS_BW_GROUP; k++) {
      if (st0 + k < nsteps) {
        const unsigned word = bits[k * 64u + lane], idx = (st0 + k) * 64u + lane;
        next[idx] = word;
        if (word) visited[idx] |= word;  // (the word is this wave's alone during the kernel)
      }
    }
    wave_sync();  // the list and the bits are reused by the next group
  }
  awake += by_head;
  awake = gdn_wave_sum(awake);
  scout = gdn_wave_sum(scout);
  const unsigned long long bh = trace ? gdn_wave_sum((unsigned long long)by_head) : 0ull;
  const unsigned long long pr = trace ? gdn_wave_sum((unsigned long long)probes) : 0ull;
  if (lane == 0) {
    s_red[wave] = awake;
    s_red[BFS_BW_WAVES + wave] = scout;
    s_red[2 * BFS_BW_WAVES + wave] = bh;
    s_red[3 * BFS_BW_WAVES + wave] = pr;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long a = 0, sc = 0, b = 0, c = 0;
    for (int i = 0; i < BFS_BW_WAVES; i++) {
      a += s_red[i];
      sc += s_red[BFS_BW_WAVES + i];
      b += s_red[2 * BFS_BW_WAVES + i];
      c += s_red[3 * BFS_BW_WAVES + i];
    }
    if (a) {
      atomicAdd(&cnt->awake, a);
      atomicAdd(&cnt->scout, sc);
    }
    if (b | c) {
      atomicAdd(&cnt->bu_by_head, b);
      atomicAdd(&cnt->bu_probes, c);
    }
  }
}

// the frontier a top-down level has just discovered, as a bitmap: the bits `visited` gained since the snapshot taken in
// front of that level (16-byte accesses; nwords is a multiple of 64).  Replaces a memset + one atomicOr per queued vertex:
// RMAT-27, 3-4.5 M vertices, 97-154 us -> 12 us + the 6 us snapshot
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_bitmap_diff_kernel(const uint4 *__restrict__ visited, const uint4 *__restrict__ snap, uint4 *__restrict__ front, unsigned nquads) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < nquads) {
    const uint4 a = visited[i], b = snap[i];
    front[i] = make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w);
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
bfs_queue_to_bitmap(const vid_t *__restrict__ q, unsigned n, unsigned *__restrict__ bits) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < n) {
    const vid_t u = q[i];
    atomicOr(&bits[u >> 5], 1u << (u & 31));
  }
}

// 8 bitmap words per thread, block scan of the popcounts, ONE atomicAdd per workgroup (a wave-level atomic per 64
// words made this conversion cost 0.76 ms on RMAT-27: 65 K atomics on one counter at ~12 ns each)
#define BFS_B2Q_WORDS 8
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_bitmap_to_queue(const unsigned *__restrict__ bits, unsigned nwords, vid_t *__restrict__ q,
                    BfsCounters *cnt, unsigned cap) {
  __shared__ unsigned s_scr[GDN_WAVES_PER_BLOCK];
  __shared__ unsigned s_base;
  const unsigned w0 = (blockIdx.x * GDN_BLOCK + threadIdx.x) * BFS_B2Q_WORDS;
  unsigned word[BFS_B2Q_WORDS];
  unsigned n = 0;
#pragma unroll
  for (int j = 0; j < BFS_B2Q_WORDS; j++) {
    word[j] = (w0 + j < nwords) ? bits[w0 + j] : 0u;
    n += __popc(word[j]);
  }
  unsigned total;
  const unsigned ex = gdn_block_excl_scan(n, s_scr, &total);
  if (total == 0) return;  // uniform
  if (threadIdx.x == 0) s_base = atomicAdd(&cnt->next_count, total);
  __syncthreads();
  unsigned pos = s_base + ex;
#pragma unroll
  for (int j = 0; j < BFS_B2Q_WORDS; j++) {
    unsigned wd = word[j];
    while (wd) {
      const int b = __ffs((int)wd) - 1;
      wd &= wd - 1u;
      if (pos < cap) q[pos] = (vid_t)((w0 + j) * 32u + (unsigned)b);
      else cnt->overflow = 1u;
      pos++;
    }
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
bfs_seed_kernel(int32_t source, int32_t *depth, unsigned *visited, vid_t *q) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    depth[source] = 0;
    visited[source >> 5] = 1u << (source & 31);
    q[0] = source;
  }
}

// sum of out-degrees of reached vertices (TEPS numerator)
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_reached_edges(const eoff_t *__restrict__ rowptr, const int32_t *__restrict__ depth, int32_t m,
                  int32_t unreached, unsigned long long *out) {
  __shared__ unsigned long long s[GDN_WAVES_PER_BLOCK];
  unsigned long long acc = 0;
  for (size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; v < (size_t)m; v += (size_t)gridDim.x * GDN_BLOCK)
    if (depth[v] != unreached) acc += rowptr[v + 1] - rowptr[v];
  acc = gdn_block_sum(acc, s);
  if (threadIdx.x == 0 && acc) atomicAdd(out, acc);
}

int gdn_reached_edges(const gdn_graph *g, const int32_t *d_dist, int32_t unreached, uint64_t *out) {
  DevBuf<unsigned long long> acc;
  GDN_TRY(acc.alloc(1));
  GDN_HIP(hipMemset(acc.p, 0, 8));
  unsigned nb = gdn_nblocks((uint64_t)g->m);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(bfs_reached_edges, dim3(nb), dim3(GDN_BLOCK), 0, 0, g->rowptr, d_dist, g->m, unreached, acc.p);
  GDN_HIP(hipGetLastError());
  unsigned long long h = 0;
  GDN_HIP(hipMemcpy(&h, acc.p, 8, hipMemcpyDeviceToHost));
  *out = h;
  return GDN_OK;
}

// ------------------------------------------------------------------------------------------
// Dense level = one propagation-blocked sweep over ALL in-edges (gdn_pb.hpp layout of the
// in-CSR, 1 bit per edge instead of a value).  It replaces the bottom-up step for the big
// levels: bfs_bu_kernel spends 21 ms per RMAT-27 level on divergent probes of the 16 MiB
// frontier bitmap when most rows are still unvisited, the sweep streams 2 B/edge twice (~2.5 ms) whatever the
// frontier is.  Once >= 3/4 of the rows that have in-edges are visited the roles flip: the bottom-up step only
// walks the few unvisited rows and stops at the first frontier parent, so the resident plan takes it for the late
// heavy levels (gdn_bfs_run).
//   phase A (per source chunk): frontier bits of the chunk -> LDS; for every group of 8 edges
//           one byte = the 8 frontier bits of their sources, stored at the group's bin-major place
//   phase B (per destination bin): visited bits of the bin -> LDS; for every non-zero byte load the
//           8 destination ids and OR the unvisited ones into the bin's new-frontier bits (LDS);
//           epilogue writes depth / visited / next frontier for the bin (coalesced by row).
// ------------------------------------------------------------------------------------------
typedef unsigned short bfs_u16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(PB_THREADS)
bfs_pb_expand_kernel(const unsigned *__restrict__ front, int log_chunk, const eoff_t *__restrict__ chunk_ptr,
                     const uint32_t *__restrict__ chunk_order, const uint16_t *__restrict__ U,
                     const uint32_t *__restrict__ G, unsigned char *__restrict__ ebits) {
  __shared__ unsigned s_f[(1 << 15) / 32 + 1];
  const unsigned words = 1u << (log_chunk - 5);
  const unsigned c = chunk_order[blockIdx.x];
  for (unsigned i = threadIdx.x; i < words; i += PB_THREADS) s_f[i] = front[(size_t)c * words + i];
  if (threadIdx.x == 0) s_f[words] = 0u;  // pad edges carry U == chunk size -> bit 0 of this word
  __syncthreads();
  const eoff_t g0 = chunk_ptr[c] >> 3, g1 = chunk_ptr[c + 1] >> 3;
  const bfs_u16x8 *U8 = reinterpret_cast<const bfs_u16x8 *>(U);
  constexpr int UNR = 4;
  for (eoff_t g = g0 + threadIdx.x; g < g1; g += UNR * PB_THREADS) {
    bfs_u16x8 u[UNR];
    unsigned d[UNR];
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t gg = g + (eoff_t)r * PB_THREADS;
      if (gg < g1) {
        u[r] = __builtin_nontemporal_load(U8 + gg);
        d[r] = __builtin_nontemporal_load(G + (gg >> 1));  // a lane owns 8 edges = half a G group
      }
    }
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t gg = g + (eoff_t)r * PB_THREADS;
      if (gg < g1) {
        unsigned byte = 0;
#define BFS_BIT(e, i) byte |= ((s_f[(unsigned)(e) >> 5] >> ((unsigned)(e)&31u)) & 1u) << (i)
        BFS_BIT(u[r].s0, 0);
        BFS_BIT(u[r].s1, 1);
        BFS_BIT(u[r].s2, 2);
        BFS_BIT(u[r].s3, 3);
        BFS_BIT(u[r].s4, 4);
        BFS_BIT(u[r].s5, 5);
        BFS_BIT(u[r].s6, 6);
        BFS_BIT(u[r].s7, 7);
#undef BFS_BIT
        ebits[2 * (size_t)d[r] + (size_t)(gg & 1)] = (unsigned char)byte;
      }
    }
  }
}

__global__ void __launch_bounds__(PB_THREADS)
bfs_pb_accumulate_kernel(int32_t m, int log_bin, const eoff_t *__restrict__ bin_ptr,
                         const uint32_t *__restrict__ bin_order, const uint16_t *__restrict__ V,
                         const unsigned char *__restrict__ ebits, unsigned *__restrict__ visited,
                         unsigned *__restrict__ next_front, int32_t *__restrict__ depth, int32_t next_level,
                         const eoff_t *__restrict__ out_rowptr, BfsCounters *cnt) {
  __shared__ unsigned s_vis[(1 << 15) / 32];
  __shared__ unsigned s_new[(1 << 15) / 32];
  __shared__ unsigned long long s_red[2 * PB_WAVES];
  const unsigned words = 1u << (log_bin - 5);
  const unsigned b = bin_order[blockIdx.x];
  const size_t w0 = (size_t)b * words;
  for (unsigned i = threadIdx.x; i < words; i += PB_THREADS) {
    s_vis[i] = visited[w0 + i];
    s_new[i] = 0u;
  }
  __syncthreads();
  const eoff_t g0 = bin_ptr[b] >> 3, g1 = bin_ptr[b + 1] >> 3;
  const bfs_u16x8 *V8 = reinterpret_cast<const bfs_u16x8 *>(V);
  // two dependent loads per group (edge byte, then the 8 row ids of a non-zero byte): all byte loads of a
  // step are issued first, then all row-id loads, so a wave keeps UNR of each in flight
  constexpr int UNR = 8;
  for (eoff_t g = g0 + threadIdx.x; g < g1; g += UNR * PB_THREADS) {
    unsigned by[UNR];
    bfs_u16x8 v[UNR];
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t gg = g + (eoff_t)r * PB_THREADS;
      by[r] = (gg < g1) ? (unsigned)__builtin_nontemporal_load(ebits + gg) : 0u;
    }
#pragma unroll
    for (int r = 0; r < UNR; r++)
      if (by[r]) v[r] = __builtin_nontemporal_load(V8 + g + (eoff_t)r * PB_THREADS);
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      if (by[r]) {
#define BFS_HIT(e, i)                                                               \
  if ((by[r] >> (i)) & 1u) {                                                        \
    const unsigned bit = 1u << ((unsigned)(e)&31u);                                 \
    if (!(s_vis[(unsigned)(e) >> 5] & bit)) atomicOr(&s_new[(unsigned)(e) >> 5], bit); \
  }
        BFS_HIT(v[r].s0, 0)
        BFS_HIT(v[r].s1, 1)
        BFS_HIT(v[r].s2, 2)
        BFS_HIT(v[r].s3, 3)
        BFS_HIT(v[r].s4, 4)
        BFS_HIT(v[r].s5, 5)
        BFS_HIT(v[r].s6, 6)
        BFS_HIT(v[r].s7, 7)
#undef BFS_HIT
      }
    }
  }
  __syncthreads();
  unsigned long long awake = 0, scout = 0;
  for (unsigned i = threadIdx.x; i < words; i += PB_THREADS) {
    const unsigned nb = s_new[i];
    next_front[w0 + i] = nb;
    if (nb) visited[w0 + i] = s_vis[i] | nb;
  }
  // one lane per ROW (a wave covers two bitmap words): the depth stores and row-offset loads of the discovered rows
  // are consecutive across the lanes (a heavy level discovers a third of all rows)
  {
    const unsigned lane = gdn_lane(), wv = threadIdx.x >> 6;
    for (unsigned i0 = wv * 2u; i0 < words; i0 += 2u * PB_WAVES) {
      const unsigned i = i0 + (lane >> 5);
      const unsigned nb = i < words ? s_new[i] : 0u;
      if ((nb >> (lane & 31u)) & 1u) {
        const size_t row = ((size_t)w0 + i) * 32 + (lane & 31u);
        if (row < (size_t)m) {
          if (depth) depth[row] = next_level;  // (null: deferred, see bfs_depth_finish_kernel)
          awake++;
          scout += out_rowptr[row + 1] - out_rowptr[row];
        }
      }
    }
  }
  awake = gdn_wave_sum(awake);
  scout = gdn_wave_sum(scout);
  const unsigned w = threadIdx.x >> 6;
  if (gdn_lane() == 0) {
    s_red[w] = awake;
    s_red[PB_WAVES + w] = scout;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long a = 0, sc = 0;
    for (int i = 0; i < PB_WAVES; i++) {
      a += s_red[i];
      sc += s_red[PB_WAVES + i];
    }
    if (a) {
      atomicAdd(&cnt->awake, a);
      atomicAdd(&cnt->scout, sc);
    }
  }
}

// ------------------------------------------------------------------------------------------
// BINNED top-down level: propagation blocking made on the fly, for a frontier that owns between ~1/256 and 1/3 of the
// edges.  The plain top-down step pays a random access per edge (the visited word) and three per discovery; the dense
// sweep streams ALL edges whatever the frontier is (RMAT-27: 2.7 ms); the bottom-up step needs the frontier to own a
// third of the edges.  Here only the frontier's out-edges move, all of it as streams:
//   bfs_btd_bin_kernel    the load-balanced expansion of the frontier (gdn_expand.hpp); a wave step's destinations are
//                         grouped by bin (dst >> logb: a ballot match), each group reserves room in its bin's list with one
//                         atomic and writes its ids side by side.  BFS_BTD_SUB lists per bin, one per XCD, keep the
//                         reservation counters off each other's cache lines and every list inside one L2.
//   bfs_btd_apply_kernel  one workgroup per bin: its lists are streamed once, the ids set bits of the bin's slice of the
//                         vertex space in LDS (2^logb bits), and the epilogue of the dense sweep follows -- new = bits &
//                         ~visited, next frontier / visited / depth / counters, all coalesced.
// 4 B read + 4 B written + 4 B read per frontier edge.  A list that would overflow (ids far from uniform over the bins) sets
// a flag, the apply kernel then does nothing, and the host runs the level on another engine.
// ------------------------------------------------------------------------------------------
#define BFS_BTD_SUB 8  // one list per XCD: a list's lines are then written through ONE L2 and leave it whole
__device__ __forceinline__ unsigned bfs_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }  // hwreg(HW_REG_XCC_ID, 0, 4)
// A list's counter is only ever added to from the CUs of ONE XCD (sub = its id), and read by the apply kernel behind a
// kernel boundary: the add can stay in that XCD's L2 (workgroup scope = no sc1) instead of travelling to the memory side
// like a device-scope atomic -- a third of its latency, and the four reservations of a work item wait for it.
__device__ __forceinline__ unsigned bfs_btd_reserve(unsigned *counter, unsigned n) {
  return __hip_atomic_fetch_add(counter, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#define BFS_BTD_MAX_LOGB 19  // 64 KB of LDS bits per bin
#define BFS_BTD_THREADS 1024
struct BfsBinVis {
  const vid_t *__restrict__ colidx;
  vid_t *__restrict__ buf;
  unsigned *cur;  // counter of list (bin, sub) at cur[(bin * SUB + sub) * 32]: a 128-byte line each
  unsigned *overflow;
  unsigned cap_each, sub;
  int logb, bin_bits;
  __device__ __forceinline__ void begin_big(vid_t) {}
  __device__ __forceinline__ void edge(int, eoff_t k, bool valid) {
    vid_t dst = 0;
    unsigned bin = 0;
    if (valid) {
      dst = __builtin_nontemporal_load(colidx + k);
      bin = (unsigned)dst >> logb;
    }
    unsigned long long peers = __ballot(valid);
    if (peers == 0ull) return;
    for (int b = 0; b < bin_bits; b++) {
      const bool one = (bin >> b) & 1u;
      const unsigned long long mk = __ballot(one && valid);
      peers &= one ? mk : ~mk;
    }
    const unsigned lane = gdn_lane();
    const unsigned rank = (unsigned)__popcll(peers & gdn_lanemask_lt());
    const size_t slot = (size_t)bin * BFS_BTD_SUB + sub;
    unsigned base = 0;
    if (valid && rank == 0u) base = bfs_btd_reserve(cur + slot * 32, (unsigned)__popcll(peers));
    base = __shfl(base, valid ? __ffsll((long long)peers) - 1 : (int)lane, 64);
    if (valid) {
      const unsigned pos = base + rank;
      if (pos < cap_each) buf[slot * cap_each + pos] = dst;
      else *overflow = 1u;
    }
  }
  __device__ __forceinline__ void finish() {}
};

__global__ void __launch_bounds__(GDN_BLOCK)
bfs_btd_bin_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ inq, unsigned nf, ExpBigList big, BfsBinVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vid_t v = 0;
  if (i < nf) {
    v = inq[i];
    b = rowptr[v];
    e = rowptr[v + 1];
  }
  vis.sub = bfs_xcc_id() & (BFS_BTD_SUB - 1);
  gdn_expand_wave(b, e, v, big, vis, s_scan[threadIdx.x >> 6]);
}

// the big-row work items (EXP_CHUNK = 256 consecutive edges of one row: four wave steps): all four steps' ids are loaded,
// matched and reserved before the first id is stored -- a step at a time, every step waited for its reservation to come
// back (a device-scope atomic under load: 1-2 us) before the next load was issued, and 278 M edges took 1.5 ms
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_btd_bin_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, BfsBinVis vis) {
  static_assert(EXP_CHUNK == 256, "four wave steps per work item");
  const unsigned lane = gdn_lane();
  const unsigned nwaves = gridDim.x * GDN_WAVES_PER_BLOCK;
  const unsigned gw = blockIdx.x * GDN_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const unsigned sub = bfs_xcc_id() & (BFS_BTD_SUB - 1);
  const unsigned long long lt = gdn_lanemask_lt();
  unsigned n = *big.count;
  if (n > big.capacity) n = big.capacity;
  // (the ids of the NEXT item are on their way while this one is matched and stored: two items' loads in flight per wave)
  vid_t nxt[4];
  bool nok[4];
  auto fetch = [&](unsigned it) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      nok[r] = false;
      nxt[r] = 0;
    }
    if (it >= n) return;
    const unsigned long long item = big.items[it];
    const vid_t v = (vid_t)(unsigned)(item & 0xFFFFFFFFull);
    const unsigned c = (unsigned)(item >> 32);
    const eoff_t rb = rowptr[v], re = rowptr[v + 1];
    const eoff_t bb = rb + (eoff_t)c * EXP_CHUNK;
    const eoff_t ee = (bb + EXP_CHUNK < re) ? bb + EXP_CHUNK : re;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const eoff_t k = bb + (eoff_t)r * 64 + lane;
      nok[r] = k < ee;
      nxt[r] = nok[r] ? __builtin_nontemporal_load(vis.colidx + k) : 0;
    }
  };
  fetch(gw);
  for (unsigned it = gw; it < n; it += nwaves) {
    vid_t dst[4];
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      dst[r] = nxt[r];
      ok[r] = nok[r];
    }
    fetch(it + nwaves);
    unsigned base[4], rank[4];
    size_t slot[4];
    int src[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const unsigned bin = (unsigned)dst[r] >> vis.logb;
      unsigned long long peers = __ballot(ok[r]);
      for (int b = 0; b < vis.bin_bits; b++) {
        const bool one = (bin >> b) & 1u;
        const unsigned long long mk = __ballot(one && ok[r]);
        peers &= one ? mk : ~mk;
      }
      rank[r] = (unsigned)__popcll(peers & lt);
      slot[r] = (size_t)bin * BFS_BTD_SUB + sub;
      src[r] = ok[r] ? __ffsll((long long)peers) - 1 : (int)lane;
      base[r] = 0;
      if (ok[r] && rank[r] == 0u) base[r] = bfs_btd_reserve(vis.cur + slot[r] * 32, (unsigned)__popcll(peers));
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const unsigned pos = __shfl(base[r], src[r], 64) + rank[r];
      if (ok[r]) {
        if (pos < vis.cap_each) vis.buf[slot[r] * vis.cap_each + pos] = dst[r];
        else *vis.overflow = 1u;
      }
    }
  }
}

// (round 4, measured and dropped: four CONSECUTIVE ids per lane -- one 16-byte load per item, the item's ids matched bin by bin
// with one reservation per distinct bin, 16-byte stores: 3.17 ms against 2.07 for the level of RMAT-27's slow source.  A hub
// row of 4 000 ids over 2^27 vertices puts 256 consecutive ids into ~16 bins, and the rounds -- one reservation each --
// run one after the other where the four steps above have theirs in flight together; profiles/sessions/r04_33.sh)
__global__ void __launch_bounds__(BFS_BTD_THREADS)
bfs_btd_apply_kernel(const vid_t *__restrict__ buf, unsigned *cur, const unsigned *__restrict__ overflow, unsigned cap_each, int logb,
                     int32_t m, unsigned nwords_pad, unsigned *__restrict__ visited, unsigned *__restrict__ next_front,
                     int32_t *__restrict__ depth, int32_t next_level, const eoff_t *__restrict__ out_rowptr, BfsCounters *cnt,
                     const unsigned long long *__restrict__ rec = nullptr /* head records: out-degrees in one load */) {
  extern __shared__ unsigned s_bits[];  // 2^(logb - 5) words
  __shared__ unsigned long long s_red[2 * (BFS_BTD_THREADS / 64)];
  const unsigned bin = blockIdx.x, words = 1u << (logb - 5);
  const bool skip = *overflow != 0u;  // the lists are incomplete: the host repeats the level on another engine
  for (unsigned i = threadIdx.x; i < words; i += BFS_BTD_THREADS) s_bits[i] = 0u;
  __syncthreads();
  const unsigned idmask = (1u << logb) - 1u;
  for (unsigned sub = 0; sub < BFS_BTD_SUB && !skip; sub++) {
    const size_t slot = (size_t)bin * BFS_BTD_SUB + sub;
    unsigned n = cur[slot * 32];
    n = n < cap_each ? n : cap_each;
    const vid_t *__restrict__ src = buf + slot * cap_each;
    // four ids per lane and load (the order inside a list means nothing; lists start on 16-byte boundaries: cap_each is a
    // multiple of 4), two loads in flight: a quarter of the vector-memory instructions of one id per lane
    typedef int bfs_i32x4 __attribute__((ext_vector_type(4)));
    const bfs_i32x4 *__restrict__ src4 = reinterpret_cast<const bfs_i32x4 *>(src);
    const unsigned n4 = n >> 2;
    constexpr int UNR = 2;
    for (unsigned i0 = threadIdx.x; i0 < n4; i0 += UNR * BFS_BTD_THREADS) {
      bfs_i32x4 id[UNR];
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const unsigned i = i0 + (unsigned)r * BFS_BTD_THREADS;
        id[r] = i < n4 ? __builtin_nontemporal_load(src4 + i) : bfs_i32x4{-1, -1, -1, -1};
      }
#pragma unroll
      for (int r = 0; r < UNR; r++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (id[r][j] >= 0) atomicOr(&s_bits[((unsigned)id[r][j] & idmask) >> 5], 1u << ((unsigned)id[r][j] & 31u));
    }
    for (unsigned i = (n4 << 2) + threadIdx.x; i < n; i += BFS_BTD_THREADS) {  // the last n % 4 ids
      const vid_t id1 = src[i];
      atomicOr(&s_bits[((unsigned)id1 & idmask) >> 5], 1u << ((unsigned)id1 & 31u));
    }
  }
  __syncthreads();
  if (threadIdx.x < BFS_BTD_SUB) cur[((size_t)bin * BFS_BTD_SUB + threadIdx.x) * 32] = 0u;  // ready for the next level
  if (skip) return;
  const size_t w0 = (size_t)bin * words;
  for (unsigned i = threadIdx.x; i < words; i += BFS_BTD_THREADS) {
    unsigned nb = 0;
    if (w0 + i < nwords_pad) {
      const unsigned vis = visited[w0 + i];
      nb = s_bits[i] & ~vis;
      next_front[w0 + i] = nb;
      if (nb) visited[w0 + i] = vis | nb;
    }
    s_bits[i] = nb;
  }
  __syncthreads();
  // one lane per ROW (a wave covers two words): depth stores and row-offset loads of the discovered rows are consecutive
  unsigned long long awake = 0, scout = 0;
  {
    const unsigned lane = gdn_lane(), wv = threadIdx.x >> 6;
    for (unsigned i0 = wv * 2u; i0 < words; i0 += 2u * (BFS_BTD_THREADS / 64)) {
      const unsigned i = i0 + (lane >> 5);
      const unsigned nb = i < words ? s_bits[i] : 0u;
      if ((nb >> (lane & 31u)) & 1u) {
        const size_t row = (w0 + i) * 32 + (lane & 31u);
        if (row < (size_t)m) {
          if (depth) depth[row] = next_level;  // (null: deferred, see bfs_depth_finish_kernel)
          awake++;
          scout += rec ? (eoff_t)BFS_REC_DEG(rec[row]) : out_rowptr[row + 1] - out_rowptr[row];
        }
      }
    }
  }
  awake = gdn_wave_sum(awake);
  scout = gdn_wave_sum(scout);
  const unsigned w = threadIdx.x >> 6;
  if (gdn_lane() == 0) {
    s_red[w] = awake;
    s_red[BFS_BTD_THREADS / 64 + w] = scout;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long a = 0, sc = 0;
    for (int i = 0; i < BFS_BTD_THREADS / 64; i++) {
      a += s_red[i];
      sc += s_red[BFS_BTD_THREADS / 64 + i];
    }
    if (a) {
      atomicAdd(&cnt->awake, a);
      atomicAdd(&cnt->scout, sc);
    }
  }
}

int gdn_radix_sort_u64(unsigned long long *a, unsigned long long *b, unsigned long long n, unsigned begin_bit, unsigned end_bit,
                       const unsigned long long **sorted);

struct gdn_bfs_plan {
  const gdn_graph *g = nullptr, *gin = nullptr;
  bool dense = false;
  PbPlan pb;  // layout of the in-CSR (no vals)
  DevBuf<unsigned char> ebits;
  DevBuf<unsigned> visited, front, next;
  DevBuf<unsigned> snap;  // visited as it was before the last top-down level (dense plans): the frontier as a bitmap = visited ^ snap
  DevBuf<vid_t> q0, q1;
  DevBuf<unsigned long long> bigitems;
  DevBuf<BfsCounters> cnt;
  DevBuf<BfsCoopCnt> coop_cnt;  // 3 rotating sets of the cooperative light-level kernel
  DevBuf<unsigned> coop_bar;    // its grid barrier (gdn_grid_barrier, gdn_common.hpp)
  int coop_blocks = 0;          // 0: cooperative launches unavailable
  DevBuf<BfsSmallOut> small_out;
  GdnMailbox mail;  // the per-level read back of the counters (gdn_common.hpp)
  unsigned nwords = 0, nwords_pad = 0, qcap = 0, bigcap = 0;
  unsigned long long active_rows = 0;  // rows with in-edges (only they can be discovered)
  DevBuf<unsigned> noin;               // bitmap of the rows without in-edges (bottom-up steps skip them)
  DevBuf<unsigned> filt;               // 2^BFS_FILT_LOG bits: hashed filter of a small frontier (bfs_filt_build_kernel)
  unsigned hub_min_deg = 0;            // out-degree of the last hub
  bool skewed = false;                 // ... >= 16 x the average degree: hub heads find most parents (bfs_plan_init)
  DevBuf<vid_t> hub_id;                // hub heads of the bottom-up step (bfs_bu_kernel): the BFS_HUBS vertices of highest
  DevBuf<unsigned long long> head;     //   out-degree, every row's out-degree | head (bfs_hub_head_kernel),
  DevBuf<unsigned> hub_front;          //   the hubs' frontier bits per level
  DevBuf<unsigned long long> head_c;   //   the head records of the rows with in-edges only (compact copy for bfs_bu_wave_kernel)
  DevBuf<unsigned> head_cbase;         //   ... and the offset of every 32-row word in it
  DevBuf<eoff_t> in_rowptr_c;          //   ... and the in-CSR offsets of the same rows
  DevBuf<unsigned> lvl;                //   BFS_DEFER_MAX frontier bitmaps kept per search (deferred depths, bfs_depth_finish_kernel)
  DevBuf<unsigned> hub_front2;         //   the same by rank for all n_ranked hubs (outer hubs: read from L2, not LDS)
  unsigned n_ranked = BFS_HUBS;        //   hubs named by rank in the head records
  // binned top-down levels (bfs_btd_*): nbins x BFS_BTD_SUB id lists of btd_cap_each entries, their counters, the flag
  DevBuf<vid_t> btd_buf;
  DevBuf<unsigned> btd_cur, btd_flag;
  unsigned btd_cap_each = 0, btd_nbins = 0;
  int btd_logb = 0, btd_bin_bits = 0;
  uint64_t btd_max_edges = 0;  // frontiers of up to this many out-edges (0: the engine is off)
  double prep_ms = 0;
};

__global__ void __launch_bounds__(GDN_BLOCK)
bfs_noin_kernel(const eoff_t *__restrict__ rowptr, int32_t m, unsigned nwords, unsigned *__restrict__ noin) {
  const unsigned w = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (w >= nwords) return;
  unsigned word = 0;
  for (unsigned i = 0; i < 32; i++) {
    const size_t v = (size_t)w * 32 + i;
    if (v >= (size_t)m || rowptr[v + 1] == rowptr[v]) word |= 1u << i;
  }
  noin[w] = word;
}

// DEFERRED depths (round 5).  A bottom-up level that discovers a third of the graph writes 4 bytes into almost every 64-byte
// line of the distance array -- scattered over the level, each line fetched, merged and written back -- although the level's
// frontier bitmap already says the same.  The wave kernel therefore leaves the distances alone (depth == nullptr) and its
// `next` bitmap is kept (a pool of BFS_DEFER_MAX bitmaps the front / next pair walks through); the search ends with ONE
// sequential pass: not visited -> unreached (this replaces the fill in front of the search), in kept bitmap k -> its level,
// visited otherwise -> written by the level that found it (the light levels write as before).
#define BFS_DEFER_MAX 8
struct BfsLevelMaps {
  const unsigned *bits[BFS_DEFER_MAX];
  int32_t level[BFS_DEFER_MAX];
  int n;
};
#define BFS_FINISH_UNR 4
// NMAPS = maps.n (compile time: the kernel is VALU-bound on choosing a level per vertex -- with loops over all BFS_DEFER_MAX slots it
// took 0.196 ms where a plain fill of the same 537 MB takes 0.078, profiles/r06_bfs_slow_source.md; a search keeps 3-5 levels)
template <int NMAPS>
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_depth_finish_kernel(const unsigned *__restrict__ visited, BfsLevelMaps maps, int32_t *__restrict__ depth, int32_t m, int32_t unreached) {
  // four consecutive vertices per thread and trip (a nibble of one bitmap word, one 16-byte store), a grid that loops (with a
  // thread per vertex the 524 288 workgroups of RMAT-27 were the cost: 0.55 ms for 512 MB), BFS_FINISH_UNR trips' words
  // requested before the first is used (one trip at a time the pass waited a round trip per 16 bytes: 0.18-0.25 ms)
  const size_t nquads = ((size_t)m + 3) / 4, stride = (size_t)gridDim.x * GDN_BLOCK;
  const bool aligned = (reinterpret_cast<size_t>(depth) & 15u) == 0;
  for (size_t q0 = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; q0 < nquads; q0 += BFS_FINISH_UNR * stride) {
    unsigned vw[BFS_FINISH_UNR], mw[BFS_FINISH_UNR][NMAPS > 0 ? NMAPS : 1];
#pragma unroll
    for (int u = 0; u < BFS_FINISH_UNR; u++) {
      const size_t q = q0 + (size_t)u * stride;
      const unsigned w = (unsigned)((q * 4) >> 5);
      const bool on = q < nquads;
#if defined(BFS_ABL) && BFS_ABL == 4  // timing-only ablation: no bitmap is read (the pass as a plain fill)
      vw[u] = 0u;
#pragma unroll
      for (int k = 0; k < NMAPS; k++) mw[u][k] = 0u;
#else
      vw[u] = on ? visited[w] : 0u;
      // (static indices: a run-time index into the by-value struct would put it into scratch memory)
#pragma unroll
      for (int k = 0; k < NMAPS; k++) mw[u][k] = on ? maps.bits[k][w] : 0u;
#endif
    }
#pragma unroll
    for (int u = 0; u < BFS_FINISH_UNR; u++) {
      const size_t q = q0 + (size_t)u * stride;
      if (q >= nquads) break;
      const size_t v0 = q * 4;
      const unsigned sh = (unsigned)(v0 & 31u);
      const unsigned seen = (vw[u] >> sh) & 15u;
      int32_t val[4] = {unreached, unreached, unreached, unreached};
      unsigned write = ~seen & 15u;
#pragma unroll
      for (int k = 0; k < NMAPS; k++) {
        const unsigned bits = (mw[u][k] >> sh) & 15u;
        write |= bits;
        const int32_t lv = maps.level[k];
#pragma unroll
        for (int j = 0; j < 4; j++) {  // (bit j as an all-ones / all-zeros mask, the level blended in: two instructions per vertex and map)
          const int32_t t = __builtin_amdgcn_sbfe((int)bits, j, 1);
          val[j] = (val[j] & ~t) | (lv & t);
        }
      }
      if (write == 15u && v0 + 4 <= (size_t)m && aligned) {
        *reinterpret_cast<int4 *>(depth + v0) = make_int4(val[0], val[1], val[2], val[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (((write >> j) & 1u) && v0 + j < (size_t)m) depth[v0 + j] = val[j];
      }
    }
  }
}

// COMPACT head records (round 5): 61 % of RMAT-27's rows have no in-edge and never reach the bottom-up step, but their 8-byte
// records sit between the others' -- a level whose open rows are most of the live ones reads all 1.07 GB of lines for 0.42 GB of
// records.  The wave kernel therefore reads a copy that holds the rows WITH in-edges only, in row order: record of row v at
// cbase[v >> 5] + popcount(the live rows of v's 32-row word below v).
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_live_counts_kernel(const unsigned *__restrict__ noin, unsigned nwords, unsigned *__restrict__ cnt) {
  const unsigned w = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (w < nwords) cnt[w] = (unsigned)__popc(~noin[w]);  // (rows past the last one count as rows without in-edges: bfs_noin_kernel)
}
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_narrow_u64_kernel(const eoff_t *__restrict__ in, unsigned n, unsigned *__restrict__ out) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < n) out[i] = (unsigned)in[i];
}
// (the in-CSR offsets of the same rows ride along: a row without in-edges has an empty range, so entry i + 1 of the compact
// offsets is also the end of live row i -- the scan of a row whose head failed reads them instead of 16 bytes of the 8 (m + 1))
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_compact_rec_kernel(const unsigned long long *__restrict__ rec, const unsigned *__restrict__ noin, const unsigned *__restrict__ cbase,
                       int32_t m, unsigned long long *__restrict__ rec_c, const eoff_t *__restrict__ in_rowptr,
                       eoff_t *__restrict__ rowptr_c, unsigned n_live) {
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v == 0u) rowptr_c[n_live] = in_rowptr[m];
  if (v >= (unsigned)m) return;
  const unsigned live = ~noin[v >> 5];
  if ((live >> (v & 31u)) & 1u) {
    const unsigned ci = cbase[v >> 5] + (unsigned)__popc(live & ((1u << (v & 31u)) - 1u));
    rec_c[ci] = rec[v];
    rowptr_c[ci] = in_rowptr[v];
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
bfs_count_rows_kernel(const eoff_t *__restrict__ rowptr, int32_t m, unsigned long long *__restrict__ out) {
  unsigned long long n = 0;
  for (size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; v < (size_t)m; v += (size_t)gridDim.x * GDN_BLOCK)
    n += rowptr[v + 1] > rowptr[v] ? 1u : 0u;
  n = gdn_wave_sum(n);
  if (gdn_lane() == 0 && n) atomicAdd(out, n);
}

// the per-level read back of the counters

static int bfs_plan_init(gdn_bfs_plan &p, const gdn_graph *g, const gdn_graph *gin, bool dense) {
  HostTimer t;
  t.start();
  p.g = g;
  p.gin = gin;
  const int32_t m = g->m;
  p.nwords = ((unsigned)m + 31u) / 32u;
  p.nwords_pad = (p.nwords + 63u) & ~63u;
  if (dense && gin) {
    // 32768-id chunks / 32768-row bins: bit slices are tiny, so take the largest tiles u16 ids allow
    int lg = 10;
    while (lg < 15 && ((int64_t)1 << (lg + 9)) < (int64_t)m) lg++;
    // tiles padded to 16 edges = 2 bytes of edge bits, one G entry per 16 edges
    GDN_TRY(pb_build(gin, m, lg, lg, p.pb, /*alloc_vals=*/false, nullptr, nullptr, false, false, /*pad=*/16, /*log_group=*/4));
    const unsigned wpad = (unsigned)((((uint64_t)(p.pb.nchunks > p.pb.nbins ? p.pb.nchunks : p.pb.nbins)) << lg) / 32u);
    if (wpad > p.nwords_pad) p.nwords_pad = wpad;
    GDN_TRY(p.ebits.alloc((p.pb.n_pad >> 3) + 8));
    GDN_HIP(hipMemset(p.ebits.p, 0, (p.pb.n_pad >> 3) + 8));
    p.dense = true;
    DevBuf<unsigned long long> nact;
    GDN_TRY(nact.alloc(1));
    GDN_HIP(hipMemset(nact.p, 0, 8));
    hipLaunchKernelGGL(bfs_count_rows_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, gin->rowptr, m, nact.p);
    GDN_HIP(hipMemcpy(&p.active_rows, nact.p, 8, hipMemcpyDeviceToHost));
    if (!(gdn_test_option("GDN_BFS_BTD") && atoi(gdn_test_option("GDN_BFS_BTD")) == 0)) {
      // binned top-down levels: about 256 bins of up to 2^19 ids (more bins beyond 2^27 vertices); room for frontiers of
      // up to a third of the edges (beyond, the bottom-up step takes the level) with a factor 2 of slack per list
      int lb = 14;
      int64_t want_bins = 256;  // measured on RMAT-27 (278 M frontier edges): 512 bins 2.19 ms, 256 bins 2.03
      if (const char *e = gdn_xoption("GDN_BFS_BTD_BINS")) want_bins = atoi(e) > 0 ? atoi(e) : want_bins;  // tuning knob
      while (lb < BFS_BTD_MAX_LOGB && (want_bins << lb) < (int64_t)m) lb++;
      p.btd_logb = lb;
      p.btd_nbins = (unsigned)(((uint64_t)m + (1ull << lb) - 1) >> lb);
      p.btd_bin_bits = 1;
      while ((1u << p.btd_bin_bits) < p.btd_nbins) p.btd_bin_bits++;
      p.btd_max_edges = g->nnz / 3 + 1;
      const uint64_t lists = (uint64_t)p.btd_nbins * BFS_BTD_SUB;
      uint64_t each = (2 * p.btd_max_edges + lists - 1) / lists;
      each = (each + 63) & ~63ull;
      if (each < 256) each = 256;
      if (each < 0x7FFFFFFFull) {
        p.btd_cap_each = (unsigned)each;
        GDN_TRY(p.btd_buf.alloc(lists * each));
        GDN_TRY(p.btd_cur.alloc(lists * 32));
        GDN_TRY(p.btd_flag.alloc(1));
        GDN_HIP(hipMemset(p.btd_cur.p, 0, lists * 32 * sizeof(unsigned)));
        GDN_HIP(hipMemset(p.btd_flag.p, 0, sizeof(unsigned)));
        if (hipFuncSetAttribute((const void *)bfs_btd_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 << (BFS_BTD_MAX_LOGB - 5)) != hipSuccess) {
          (void)hipGetLastError();
          p.btd_max_edges = 0;
        }
      } else {
        p.btd_max_edges = 0;
      }
    }
  }
  p.qcap = (unsigned)m;
  const uint64_t bigcap64 = g->nnz / EXP_CHUNK + (uint64_t)m / 64 + 1024;
  p.bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
  GDN_TRY(p.visited.alloc(p.nwords_pad));
  GDN_TRY(p.q0.alloc(p.qcap));
  GDN_TRY(p.q1.alloc(p.qcap));
  GDN_TRY(p.bigitems.alloc(p.bigcap));
  GDN_TRY(p.cnt.alloc(1));
  GDN_TRY(p.small_out.alloc(1));
  {  // the cooperative light-level kernel: one workgroup per CU if the device takes cooperative launches
    int dev = 0, coop = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) == hipSuccess &&
        coop && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bfs_td_coop_kernel, BFS_COOP_THREADS, 0) == hipSuccess && per_cu >= 1) {
      p.coop_blocks = cus;
      GDN_TRY(p.coop_cnt.alloc(3));
      GDN_TRY(p.coop_bar.alloc(GDN_GBAR_WORDS));
    }
    (void)hipGetLastError();
  }
  p.mail.init();
  if (gin) {
    GDN_TRY(p.front.alloc(p.nwords_pad));
    GDN_TRY(p.next.alloc(p.nwords_pad));
    GDN_TRY(p.noin.alloc(p.nwords_pad));
    if (dense) GDN_TRY(p.snap.alloc(p.nwords_pad));
    hipLaunchKernelGGL(bfs_noin_kernel, dim3(gdn_nblocks(p.nwords_pad)), dim3(GDN_BLOCK), 0, 0, gin->rowptr, m, p.nwords_pad,
                       p.noin.p);
    // hub heads: from 2^24 edges on (below, a level is a few hundred microseconds and the plan build should stay short);
    // GDN_BFS_HUB_HEADS=0 switches them off (A/B)
    const char *hh = gdn_xoption("GDN_BFS_HUB_HEADS");
    unsigned long long heads_from = 1ull << 24;
    if (const char *e = gdn_test_option("GDN_BFS_HEADS_MIN_NNZ")) heads_from = strtoull(e, nullptr, 10);  // (tests)
    // (a graph with fewer vertices than a few times the hub slots gains nothing; the test knob lifts that too)
    const bool enough = (unsigned)m >= 4u * BFS_HUBS || gdn_test_option("GDN_BFS_HEADS_MIN_NNZ") != nullptr;
    if (dense && g->nnz >= heads_from && enough && m >= 2 && !(hh && hh[0] == '0')) {
      HostTimer th;
      GDN_HIP(hipDeviceSynchronize());
      th.start();
      DevBuf<unsigned long long> ka, kb;
      DevBuf<unsigned> hub_idx;
      GDN_TRY(ka.alloc((size_t)m));
      GDN_TRY(kb.alloc((size_t)m));
      GDN_TRY(hub_idx.alloc((size_t)m));
      // outer hubs where the vertex-indexed frontier bitmap is beyond an XCD's L2 (GDN_BFS_HUBS2=0: without)
      {
        const char *e2 = gdn_test_option("GDN_BFS_HUBS2");
        // (default from 2^27 vertices on: RMAT-27 -1.5 %, RMAT-26 +1..2 % -- the 2^21 probes that fill the rank bitmap per level)
        p.n_ranked = (e2 ? e2[0] != '0' : (unsigned)m >= (1u << 27)) ? BFS_HUBS2 : BFS_HUBS;  // (=1 on a small graph: every head by rank)
      }
      GDN_TRY(p.hub_id.alloc(p.n_ranked));
      if (p.n_ranked > BFS_HUBS) GDN_TRY(p.hub_front2.alloc(p.n_ranked / 32));
      GDN_TRY(p.head.alloc((size_t)m));
      GDN_TRY(p.hub_front.alloc(BFS_HUBS / 32 + 2));  // + the count of hubs in the frontier (64 bits)
      GDN_TRY(p.filt.alloc((size_t)1 << (BFS_FILT_LOG - 5)));
      hipLaunchKernelGGL(bfs_hub_keys_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, ka.p);
      GDN_HIP(hipGetLastError());
      const unsigned long long *sorted = nullptr;
      GDN_TRY(gdn_radix_sort_u64(ka.p, kb.p, (unsigned long long)m, 32u, 64u, &sorted));
      hipLaunchKernelGGL(bfs_hub_rank_kernel, dim3(gdn_nblocks((uint64_t)m > p.n_ranked ? (uint64_t)m : (uint64_t)p.n_ranked)), dim3(GDN_BLOCK), 0, 0, sorted, m, p.n_ranked, p.hub_id.p,
                         hub_idx.p);
      hipLaunchKernelGGL(bfs_hub_head_kernel, dim3(256 * 16), dim3(GDN_BLOCK), 0, 0, gin->rowptr, gin->colidx, g->rowptr, m,
                         hub_idx.p, sorted, p.head.p, p.n_ranked);
      GDN_HIP(hipGetLastError());
      GDN_HIP(hipDeviceSynchronize());
      {  // deferred depths (GDN_BFS_DEFER_DEPTH=0: without; default from 2^25 vertices on: at RMAT-24 the pass at the end costs what the levels save)
        const char *ed = gdn_test_option("GDN_BFS_DEFER_DEPTH");
        if (ed ? ed[0] != '0' : (unsigned)m >= (1u << 25)) GDN_TRY(p.lvl.alloc((size_t)BFS_DEFER_MAX * p.nwords_pad));
      }
      {  // the compact copy (GDN_BFS_REC_COMPACT=0: without; default from 2^25 vertices on: RMAT-24 measures the same with and without)
        const char *ec = gdn_test_option("GDN_BFS_REC_COMPACT");
        if (ec ? ec[0] != '0' : (unsigned)m >= (1u << 25)) {
          DevBuf<unsigned> cnt;
          DevBuf<eoff_t> scan;
          GDN_TRY(cnt.alloc_scratch(p.nwords_pad));
          GDN_TRY(scan.alloc_scratch((size_t)p.nwords_pad + 1));
          hipLaunchKernelGGL(bfs_live_counts_kernel, dim3(gdn_nblocks(p.nwords_pad)), dim3(GDN_BLOCK), 0, 0, p.noin.p, p.nwords_pad, cnt.p);
          GDN_TRY(gdn_exclusive_scan_u32_to_u64(cnt.p, scan.p, (size_t)p.nwords_pad, 0));
          eoff_t n_live = 0;
          GDN_HIP(hipMemcpy(&n_live, scan.p + p.nwords_pad, sizeof(eoff_t), hipMemcpyDeviceToHost));
          GDN_TRY(p.head_cbase.alloc((size_t)p.nwords_pad + 1));
          GDN_TRY(p.head_c.alloc((size_t)n_live + 1));
          GDN_TRY(p.in_rowptr_c.alloc((size_t)n_live + 1));
          hipLaunchKernelGGL(bfs_narrow_u64_kernel, dim3(gdn_nblocks((uint64_t)p.nwords_pad + 1)), dim3(GDN_BLOCK), 0, 0, scan.p,
                             p.nwords_pad + 1u, p.head_cbase.p);
          hipLaunchKernelGGL(bfs_compact_rec_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0,
                             (const unsigned long long *)p.head.p, p.noin.p, p.head_cbase.p, m, p.head_c.p, gin->rowptr,
                             p.in_rowptr_c.p, (unsigned)n_live);
          GDN_HIP(hipGetLastError());
          GDN_HIP(hipDeviceSynchronize());
        }
      }
      // how skewed the out-degrees are: the smallest degree among the hubs against the average (R-MAT-27: hundreds against 16;
      // a uniform random graph: 35 against 16).  With heads that are real hubs the bottom-up step pays from an edge share of
      // 1/8 on (bfs_run), without it is the worst engine there.
      if ((unsigned)m > BFS_HUBS) {
        unsigned long long key = 0;
        GDN_HIP(hipMemcpy(&key, sorted + ((size_t)m - BFS_HUBS), sizeof(key), hipMemcpyDeviceToHost));
        p.hub_min_deg = (unsigned)(key >> 32);
        p.skewed = (unsigned long long)p.hub_min_deg * (unsigned long long)m >= 16ull * g->nnz;
      }
      if (gdn_option("GDN_BFS_TRACE"))
        fprintf(stderr, "[bfs] plan: heads of the bottom-up step %.1f ms; the last of the %u hubs has out-degree %u (average %.1f): %s\n",
                th.stop_ms(), BFS_HUBS, p.hub_min_deg, (double)g->nnz / (double)m, p.skewed ? "skewed" : "not skewed");
    }
  }
  GDN_HIP(hipDeviceSynchronize());
  p.prep_ms = t.stop_ms();
  return GDN_OK;
}

static int bfs_run(gdn_bfs_plan &p, int32_t source, int32_t *d_dist, gdn_stats *stats) {
  const gdn_graph *g = p.g, *gin = p.gin;
  const int32_t m = g->m;
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  st.prep_ms = p.prep_ms;
  HostTimer tsolve;
  // ---- timed region == omp_beamer.cc:128-148 PLUS the per-search initialisation, which every BFSSolver of the reference does in
  // front of its Timer (the caller's distances(m, MYINFINITY), src/bfs/main.cc:21; linear_base.cu:50-63, omp_beamer.cc:119-134):
  // solve_ms is a superset of what the reference times.  GDN_BFS_TIME_INIT=1 (bench.py's note) brackets the initialisation
  // with events and reports it in stats.prep_ms INSTEAD of the plan's build time, and the closing pass that writes the deferred
  // distances in stats.last_error (ms; NOT initialisation, see below); solve_ms stays the whole region.
  const char *ti = gdn_option("GDN_BFS_TIME_INIT");
  const bool time_init = ti && ti[0] == '1';
  struct EvSet {
    hipEvent_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
    ~EvSet() {
      for (hipEvent_t e : {a, b, c, d})
        if (e) (void)hipEventDestroy(e);
    }
  } ev;
  hipEvent_t &ev_a = ev.a, &ev_b = ev.b;
  if (time_init && (hipEventCreate(&ev_a) != hipSuccess || hipEventCreate(&ev_b) != hipSuccess || hipEventCreate(&ev.c) != hipSuccess ||
                    hipEventCreate(&ev.d) != hipSuccess)) {
    gdn_set_error("gdn_bfs: hipEventCreate failed");
    return GDN_ERR_HIP;
  }
  tsolve.start();
  // (the source's out-degree first: the read blocks, and behind the launches below it would wait for the 512 MB fill)
  eoff_t srow[2];
  GDN_HIP(hipMemcpy(srow, g->rowptr + source, sizeof(srow), hipMemcpyDeviceToHost));
  if (time_init) GDN_HIP(hipEventRecord(ev_a, 0));
  const bool defer = p.lvl.p != nullptr && p.head.p != nullptr;  // (the pass at the end writes "unreached" too: no fill)
  BfsLevelMaps maps;
  maps.n = 0;
  if (!defer) GDN_TRY(gdn_fill_i32(d_dist, GDN_MYINFINITY, (size_t)m, 0));
  GDN_HIP(hipMemsetAsync(p.visited.p, 0, (size_t)p.nwords_pad * 4, 0));
  hipLaunchKernelGGL(bfs_seed_kernel, dim3(1), dim3(64), 0, 0, source, d_dist, p.visited.p, p.q0.p);
  if (time_init) GDN_HIP(hipEventRecord(ev_b, 0));

  const int alpha = 15, beta = 18;   // omp_beamer.cc:111
  int alpha_dense = 32;              // a dense sweep costs about nnz/32 top-down edge visits
  if (const char *e = gdn_option("GDN_BFS_ALPHA_DENSE")) alpha_dense = atoi(e) > 0 ? atoi(e) : 32;  // tuning knob
  vid_t *qin = p.q0.p, *qout = p.q1.p;
  unsigned nf = 1;
  int64_t edges_to_check = (int64_t)g->nnz;
  int64_t scout_count = (int64_t)(srow[1] - srow[0]);
  int32_t level = 0;  // depth of the vertices in the current frontier
  int iter = 0;
  int64_t visited_total = 1;  // discovered so far (the source included)
  int64_t bu_frac = 2;        // bottom-up engine once <= 1/bu_frac of the rows with in-edges are undiscovered (measured on
                              // RMAT-22..27: 2 beats 4 wherever a second heavy level follows the first, 1 loses)
  if (const char *e = gdn_xoption("GDN_BFS_BU_FRAC")) bu_frac = atoi(e) > 0 ? atoi(e) : (int64_t)1 << 40;  // tuning knob (0 = never)
  // a late level stays on the bottom-up engine while its frontier still scouts more than m / bu_stay edges: the step
  // costs a scan of two bitmaps plus the few undiscovered rows, a top-down step costs two divergent row-offset reads per
  // frontier vertex (RMAT-27: 4.8 M frontier vertices that discover 28 K = 0.48 ms top-down)
  // binned top-down levels (bfs_btd_*): frontiers from nnz / alpha_btd edges (and btd_min_edges) up to a third of all
  // (GDN_BFS_BTD=2 forces them wherever they apply -- tests; GDN_BFS_ALPHA_BTD lets the bitmap phase start earlier for them:
  // a 10 M-edge frontier of RMAT-27 took 0.60 ms binned, 0.53 ms plain top-down, hence the default = alpha_dense)
  uint64_t alpha_btd = (uint64_t)alpha_dense;
  int64_t btd_min_edges = 1 << 22;
  bool btd_on = true, btd_force = false;
  if (const char *e = gdn_test_option("GDN_BFS_ALPHA_BTD")) alpha_btd = atoi(e) > 0 ? (uint64_t)atoi(e) : alpha_btd;  // tuning knobs
  if (const char *e = gdn_test_option("GDN_BFS_BTD_MIN")) btd_min_edges = atoll(e);
  if (const char *e = gdn_test_option("GDN_BFS_BTD")) {
    btd_on = atoi(e) != 0;
    btd_force = atoi(e) == 2;
  }
  // see the engine choice of a heavy level below (0 = off).  Round 5: 8 on a graph with real hubs (gdn_bfs_plan::skewed) -- with
  // head records, lone heads and the wave kernel the bottom-up step takes RMAT-27's 0.13-share level (source 5) in 1.87 ms where
  // the binned level takes 2.05 (round 4, before those: 2.45; profiles/r05_bfs_bu_scan.txt); on a uniform random graph a step
  // from 1/8 on costs 4.9 ms against 3.2 (profiles/r05_bfs_uniform26_trace.txt): there it stays at Beamer's 1/3
  int64_t bu_edge_div = p.skewed && p.head.p ? 8 : 3;
  if (const char *e = gdn_test_option("GDN_BFS_BU_EDGE_DIV")) bu_edge_div = atoi(e);  // tuning knob
  int64_t bu_stay = 256;
  if (const char *e = gdn_xoption("GDN_BFS_BU_STAY")) bu_stay = atoi(e) > 0 ? atoi(e) : bu_stay;  // tuning knob
  // in-neighbours per round trip of the bottom-up scan: 4 measured the same as 1 on RMAT-24 / 27 (session r05_08: the step is
  // not bound by the lane-private scan loops) -- the knob stays for the next graph family
  // (round 6: 0 = the FLAT scan, all in-edges of 64 open rows as one list -- see bfs_bu_wave_kernel's second stage; the lane-private
  // loops stay behind the knob: 1 = an in-neighbour per round trip, 4 = four)
  int bu_scan = 0;
  if (const char *e = gdn_xoption("GDN_BFS_BU_SCAN")) bu_scan = atoi(e) > 1 ? 4 : atoi(e) == 1 ? 1 : 0;  // A/B knob
  // largest frontier that gets the hashed filter (GDN_BFS_BU_FILTER=<vertices>, 0 = never) -- on graphs whose frontier bitmap
  // is beyond an XCD's L2 (from 2^26 vertices = 8 MB on).  Measured (profiles/r05_bfs_bu_scan.txt): RMAT-27's hub-frontier level
  // 1.865 -> 1.815 ms (source 5: 2.76 -> 2.69 ms), the other searches unchanged; RMAT-24, whose 2 MB bitmap IS L2 resident, pays
  // the 10 us of the filter's build for nothing (+1.5 %) -- the probes are a small part of that level, its gathers of the failing
  // rows' offsets and neighbour lists are the rest
  int64_t filt_max = m >= (1 << 26) ? (1 << 20) : 0;
  if (const char *e = gdn_xoption("GDN_BFS_BU_FILTER")) filt_max = atoll(e);
  unsigned hub_min = BFS_HUBS / 8;  // hubs a frontier must hold for the bottom-up step to read the heads
  if (const char *e = gdn_test_option("GDN_BFS_HUB_MIN")) hub_min = (unsigned)atoi(e);  // tuning knob
  // frontiers of at most small_nf vertices and small_scout out-edges run fused in one workgroup (0 = never)
  // (measured: 1024 / 16384 made RMAT-20..24 searches 5-15 % slower -- 16 K edges on ONE CU are no faster than a launch
  // over all of them --, a 100 000-vertex chain 4.6x faster; the smaller limits keep the second without the first)
  unsigned small_nf = 256;
  unsigned long long small_scout = 2048;
  if (const char *e = gdn_test_option("GDN_BFS_SMALL_NF")) small_nf = (unsigned)atoi(e);                  // tuning knobs
  if (const char *e = gdn_test_option("GDN_BFS_SMALL_SCOUT")) small_scout = strtoull(e, nullptr, 10);
  // light levels that outgrow the one workgroup go to the cooperative grid (bfs_td_coop_kernel) once `coop_streak` light
  // levels in a row say "high diameter" (an R-MAT search has 2-3 light levels on either side of its heavy ones and never
  // gets there; GDN_BFS_COOP=0 switches the path off, =1 takes it from the first light level: tests)
  unsigned coop_nf = 65536, coop_streak = 8;
  unsigned long long coop_scout = 1ull << 20;
  if (const char *e = gdn_option("GDN_BFS_COOP")) {
    if (atoi(e) == 0) coop_nf = 0;
    else coop_streak = 0;
  }
  if (p.coop_blocks == 0) coop_nf = 0;
  unsigned light_streak = 0;
  BfsCounters h;
  memset(&h, 0, sizeof(h));
  ExpBigList big;
  big.items = p.bigitems.p;
  big.capacity = p.bigcap;
  int32_t snap_level = -1;  // the level whose discoveries are visited ^ snap (-1: no snapshot)
  unsigned *kept_front = nullptr;  // deferred depths: that difference, already made (a kept bitmap) ...
  int32_t kept_front_level = -1;   // ... for this level
  // (levels of fewer frontier edges discover so little that the old conversion is as cheap as the snapshot)
  int64_t snap_min_edges = 1 << 16;
  if (const char *e = gdn_test_option("GDN_BFS_SNAP_MIN")) snap_min_edges = atoll(e);  // (test knob: small graphs reach the snapshot / td_keep path)
  int64_t td_defer_min = 1 << 20;  // frontier edges from which a top-down level defers its depths (a 16 MB bitmap pass against its writes)
  if (const char *e = gdn_test_option("GDN_BFS_TD_DEFER_MIN")) td_defer_min = atoll(e);  // (tuning knob; huge = never)
  // The level counters are ZEROED by the kernel that reads them back (GdnMailbox::read, zero = true): a level that follows a
  // read-back needs no memset of its own -- one dispatch per level less (11 fills per RMAT-27 search before).  cnt_reset() is
  // what a level calls before its kernels add to the counters: a memset only when the last thing that touched them was not a read.
  bool cnt_clean = false;
  auto cnt_reset = [&]() -> int {
    if (!cnt_clean) GDN_HIP(hipMemsetAsync(p.cnt.p, 0, sizeof(BfsCounters), 0));
    cnt_clean = false;  // (the caller's kernels write them next)
    return GDN_OK;
  };
  auto cnt_read = [&](BfsCounters &hc) -> int {
    GDN_TRY(p.mail.read(p.cnt.p, hc, 0, true));
    cnt_clean = true;
    return GDN_OK;
  };
  const bool trace = gdn_option("GDN_BFS_TRACE") != nullptr;  // per-level timing to stderr (adds syncs)
  const bool b2q_read = gdn_xoption("GDN_BFS_B2Q_READ") != nullptr;  // (A/B knob: read the length of a listed frontier back as before)
  HostTimer tl;
  auto lap = [&](const char *what, long long a, long long b) {
    if (trace) fprintf(stderr, "[bfs] level %d %-10s nf/awake=%lld scout=%lld  %.3f ms\n", level, what, a, b, tl.stop_ms());
    if (trace) tl.start();
  };
  if (trace) {
    (void)hipDeviceSynchronize();
    tl.start();
  }
  while (nf > 0) {
    // the bitmap engines take a level from nnz / alpha_dense frontier edges on; with the binned top-down level (which
    // moves only the frontier's edges) already from nnz / alpha_btd on
    const int64_t heavy_from = (int64_t)(g->nnz / (uint64_t)(p.btd_max_edges && btd_on ? alpha_btd : alpha_dense));
    if (p.dense && scout_count > heavy_from && (scout_count > (int64_t)(g->nnz / alpha_dense) || scout_count >= btd_min_edges)) {
      // ---- dense phase: bitmap levels while the frontier stays heavy
      light_streak = 0;
      unsigned *fr = p.front.p, *nx = p.next.p;
      if (kept_front_level == level && kept_front) {  // the top-down level before left its depths to the end: its bitmap exists
        fr = kept_front;
      } else if (snap_level == level && p.snap.p) {  // the level before was a top-down one with a snapshot in front of it
        hipLaunchKernelGGL(bfs_bitmap_diff_kernel, dim3(gdn_nblocks(p.nwords_pad / 4)), dim3(GDN_BLOCK), 0, 0,
                           reinterpret_cast<const uint4 *>(p.visited.p), reinterpret_cast<const uint4 *>(p.snap.p),
                           reinterpret_cast<uint4 *>(p.front.p), p.nwords_pad / 4);
      } else {
        GDN_HIP(hipMemsetAsync(p.front.p, 0, (size_t)p.nwords_pad * 4, 0));
        hipLaunchKernelGGL(bfs_queue_to_bitmap, dim3(gdn_nblocks(nf)), dim3(GDN_BLOCK), 0, 0, qin, nf, p.front.p);
      }
      unsigned *const scratch[2] = {p.front.p, p.next.p};
      int64_t awake = 0;
      bool have_queue = true;  // qin / nfq hold the frontier as a vertex list (what the binned level expands)
      unsigned nfq = nf;
      do {
        ++iter;
        GDN_TRY(cnt_reset());
        // engine of this heavy level: the sweep over all in-edges, or -- once few rows are left to discover --
        // the bottom-up step over the unvisited rows (omp_beamer.cc:13-31)
        // the bottom-up step also wins while MANY rows are left when the frontier owns a large share p of all edges: an
        // undiscovered row then finds a parent after ~1/p probes (Beamer's own criterion, omp_beamer.cc:130).  Measured on
        // RMAT-27: p = 0.72 bottom-up 1.70 ms against the sweep's 2.80; p = 0.13 4.11 against 2.73 -- the costs cross
        // near p = 1/4, the switch sits at 1/3
        const int64_t left = (int64_t)p.active_rows - visited_total;
        const bool bottom_up = left * bu_frac <= (int64_t)p.active_rows || (bu_edge_div > 0 && scout_count * bu_edge_div >= (int64_t)g->nnz);
        // below that share: the binned top-down level (only the frontier's edges move) instead of the sweep over all
        // (measured, RMAT-22 .. 28: the sweep costs ~1.3 ps per edge of the GRAPH, the binned level ~0.35 ms + 6 ps per
        // edge of the FRONTIER -- it pays from about a billion edges on: RMAT-27 p = 0.13 2.73 -> 2.03 ms, RMAT-28 two of
        // three sources 10.2 -> 7.6 / 8.7 ms; below, it loses: RMAT-22 0.36 -> 0.7 ms when forced)
        bool btd = !bottom_up && btd_on && p.btd_max_edges && (uint64_t)scout_count <= p.btd_max_edges &&
                   (btd_force || 100 * scout_count + 5800000000ll < 21 * (int64_t)g->nnz);
        const char *engine = bottom_up ? "bottom-up" : "dense";
        // deferred depths: this level's `next` bitmap goes into the pool and stays there, its kernels leave the distances alone
        // (every engine of the dense phase but the window form of the bottom-up step, an A/B knob)
        const char *bfe = gdn_xoption("GDN_BFS_BU_FORM");  // window: the workgroup-per-window form (bfs_bu_kernel)
        const bool window_form = bfe && bfe[0] == 'w';
        const bool keep = defer && maps.n < BFS_DEFER_MAX && !window_form;
        if (keep) {
          nx = p.lvl.p + (size_t)maps.n * p.nwords_pad;
          maps.bits[maps.n] = nx;
          maps.level[maps.n] = level + 1;
          maps.n++;
        }
        int32_t *const d_out = keep ? (int32_t *)nullptr : d_dist;
        if (btd) {
          if (!have_queue) {  // the frontier exists as a bitmap only: list it
            hipLaunchKernelGGL(bfs_bitmap_to_queue, dim3(gdn_nblocks(p.nwords, GDN_BLOCK * BFS_B2Q_WORDS)), dim3(GDN_BLOCK), 0, 0, fr,
                               p.nwords, qin, p.cnt.p, p.qcap);
            // (the list is as long as the level before discovered rows: no read-back unless it may not fit)
            if (awake > 0 && (uint64_t)awake <= (uint64_t)p.qcap && !b2q_read) {
              nfq = (unsigned)awake;
            } else {
              GDN_TRY(cnt_read(h));
              nfq = h.next_count;
            }
            have_queue = true;
            GDN_TRY(cnt_reset());
          }
          ExpBigList bbig;
          bbig.items = p.bigitems.p;
          bbig.capacity = p.bigcap;
          bbig.count = &p.cnt.p->big_count;
          bbig.overflow = &p.cnt.p->overflow;
          BfsBinVis bv;
          bv.colidx = g->colidx;
          bv.buf = p.btd_buf.p;
          bv.cur = p.btd_cur.p;
          bv.overflow = p.btd_flag.p;
          bv.cap_each = p.btd_cap_each;
          bv.sub = 0;
          bv.logb = p.btd_logb;
          bv.bin_bits = p.btd_bin_bits;
          hipLaunchKernelGGL(bfs_btd_bin_kernel, dim3(gdn_nblocks(nfq)), dim3(GDN_BLOCK), 0, 0, g->rowptr, qin, nfq, bbig, bv);
          hipLaunchKernelGGL(bfs_btd_bin_big_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, g->rowptr, bbig, bv);
          hipLaunchKernelGGL(bfs_btd_apply_kernel, dim3(p.btd_nbins), dim3(BFS_BTD_THREADS), (size_t)4 << (p.btd_logb - 5), 0,
                             p.btd_buf.p, p.btd_cur.p, p.btd_flag.p, p.btd_cap_each, p.btd_logb, m, p.nwords_pad, p.visited.p, nx, d_out,
                             level + 1, g->rowptr, p.cnt.p, (const unsigned long long *)p.head.p);
          unsigned flag = 0;
          GDN_HIP(hipMemcpy(&flag, p.btd_flag.p, sizeof(flag), hipMemcpyDeviceToHost));
          if (flag) {  // a bin's list was too short for this frontier: nothing was applied; the sweep takes the level,
            // and this search stays away from the binned engine
            GDN_HIP(hipMemsetAsync(p.btd_flag.p, 0, sizeof(unsigned), 0));
            GDN_TRY(cnt_reset());
            btd = false;
            btd_on = false;
          } else {
            engine = "binned";
          }
        }
        if (btd) {
        } else if (bottom_up) {
          if (p.head.p) {
            hipLaunchKernelGGL(bfs_hub_front_kernel, dim3(p.n_ranked / GDN_BLOCK), dim3(GDN_BLOCK), 0, 0, p.hub_id.p, fr, p.hub_front.p,
                               p.hub_front2.p);
          }
          if (p.head.p && !window_form) {
            // frontier of at most 2^20 vertices (its size is the last level's discoveries): the hashed filter in front of the bitmap
            const int64_t front_n = have_queue ? (int64_t)nfq : awake;
            const unsigned *filt = nullptr;
            if (p.filt.p && filt_max > 0 && front_n <= filt_max) {
              GDN_HIP(hipMemsetAsync(p.filt.p, 0, (size_t)4 << (BFS_FILT_LOG - 5), 0));
              hipLaunchKernelGGL(bfs_filt_build_kernel, dim3(gdn_nblocks(p.nwords_pad)), dim3(GDN_BLOCK), 0, 0, fr, p.nwords_pad, p.filt.p);
              filt = p.filt.p;
            }
            hipLaunchKernelGGL(bfs_bu_wave_kernel, dim3(BFS_BW_GRID), dim3(BFS_BW_THREADS), 0, 0, gin->rowptr, gin->colidx, m,
                               p.nwords_pad * 32u, fr, nx, p.visited.p, d_out, level + 1, p.cnt.p, p.noin.p,
                               p.head_c.p ? p.head_c.p : p.head.p, p.hub_front.p, hub_min, trace, bu_scan, filt, p.hub_front2.p,
                               p.head_cbase.p, p.in_rowptr_c.p);
          }
          else
          hipLaunchKernelGGL(bfs_bu_kernel, dim3(BFS_BU_GRID), dim3(GDN_BLOCK), 0, 0, gin->rowptr, gin->colidx, g->rowptr, m,
                             p.nwords_pad * 32u, fr, nx, p.visited.p, d_dist, level + 1, p.cnt.p, p.noin.p, p.head.p,
                             p.hub_front.p, hub_min, trace, p.hub_front2.p);
        } else {
          hipLaunchKernelGGL(bfs_pb_expand_kernel, dim3(p.pb.nchunks), dim3(PB_THREADS), 0, 0, fr, p.pb.log_chunk,
                             p.pb.chunk_ptr.p, p.pb.chunk_order.p, p.pb.U.p, p.pb.G.p, p.ebits.p);
          hipLaunchKernelGGL(bfs_pb_accumulate_kernel, dim3(p.pb.nbins), dim3(PB_THREADS), 0, 0, m, p.pb.log_bin,
                             p.pb.bin_ptr.p, p.pb.bin_order.p, p.pb.V.p, p.ebits.p, p.visited.p, nx, d_out, level + 1,
                             g->rowptr, p.cnt.p);
        }
        GDN_TRY(cnt_read(h));
        if (h.overflow) {
          gdn_set_error("gdn_bfs: device worklist overflow");
          return GDN_ERR_OVERFLOW;
        }
        awake = (int64_t)h.awake;
        scout_count = (int64_t)h.scout;
        visited_total += awake;
        fr = nx;  // (a kept bitmap is never written again: the next level writes into a scratch one that is not its frontier)
        nx = fr == scratch[0] ? scratch[1] : scratch[0];
        level++;
        have_queue = false;
        if (trace && bottom_up && !btd)
        {
          unsigned nh = 0;
          if (p.head.p) {
            std::vector<unsigned> hw(BFS_HUBS / 32);
            (void)hipMemcpy(hw.data(), p.hub_front.p, hw.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
            for (unsigned x : hw) nh += (unsigned)__builtin_popcount(x);
          }
          fprintf(stderr, "[bfs]   bottom-up: %u hubs in the frontier, %llu rows by their hub head, %llu in-neighbours probed for the others\n",
                  nh, h.bu_by_head, h.bu_probes);
        }
        lap(engine, awake, scout_count);
        // stay on bitmaps while the frontier is heavy, or while the cheap bottom-up engine beats a top-down
        // step over scout_count edges (a late level with millions of frontier vertices but few discoveries)
      } while (awake > 0 && (scout_count > (int64_t)(g->nnz / alpha_dense) ||
                             (btd_on && p.btd_max_edges && scout_count > (int64_t)(g->nnz / alpha_btd) && scout_count >= btd_min_edges) ||
                             (((int64_t)p.active_rows - visited_total) * bu_frac <= (int64_t)p.active_rows &&
                              scout_count > (int64_t)m / bu_stay)));
      if (awake == 0) {
        nf = 0;
        break;
      }
      GDN_TRY(cnt_reset());
      hipLaunchKernelGGL(bfs_bitmap_to_queue, dim3(gdn_nblocks(p.nwords, GDN_BLOCK * BFS_B2Q_WORDS)), dim3(GDN_BLOCK), 0, 0, fr, p.nwords, qin,
                         p.cnt.p, p.qcap);
      // the frontier is what the last level discovered: its size is known, the read-back of the list's length (a round trip of
      // ~25 us: 2 % of an RMAT-27 search, 6 % at RMAT-24) only where the list may not fit
      if ((uint64_t)awake <= (uint64_t)p.qcap && !b2q_read) {
        nf = (unsigned)awake;
      } else {
        GDN_TRY(cnt_read(h));
        nf = h.next_count;
      }
      lap("bitmap2q", nf, 0);
      edges_to_check = 0;  // from here on only the top-down tail is left
    } else if (!p.dense && gin != nullptr && scout_count > edges_to_check / alpha) {
      // ---- bottom-up phase (omp_beamer.cc:130-141)
      light_streak = 0;
      GDN_HIP(hipMemsetAsync(p.front.p, 0, (size_t)p.nwords_pad * 4, 0));
      hipLaunchKernelGGL(bfs_queue_to_bitmap, dim3(gdn_nblocks(nf)), dim3(GDN_BLOCK), 0, 0, qin, nf, p.front.p);
      int64_t awake = (int64_t)nf, old_awake;
      unsigned *fr = p.front.p, *nx = p.next.p;
      do {
        ++iter;
        old_awake = awake;
        GDN_TRY(cnt_reset());
        hipLaunchKernelGGL(bfs_bu_kernel, dim3(BFS_BU_GRID), dim3(GDN_BLOCK), 0, 0, gin->rowptr, gin->colidx, g->rowptr, m,
                           p.nwords_pad * 32u, fr, nx, p.visited.p, d_dist, level + 1, p.cnt.p, p.noin.p);
        GDN_TRY(cnt_read(h));
        awake = (int64_t)h.awake;
        unsigned *t = fr;
        fr = nx;
        nx = t;
        level++;
      } while (awake >= old_awake || awake > m / beta);
      GDN_TRY(cnt_reset());
      hipLaunchKernelGGL(bfs_bitmap_to_queue, dim3(gdn_nblocks(p.nwords, GDN_BLOCK * BFS_B2Q_WORDS)), dim3(GDN_BLOCK), 0, 0, fr, p.nwords, qin,
                         p.cnt.p, p.qcap);
      GDN_TRY(cnt_read(h));
      nf = h.next_count;
      scout_count = 1;
    } else {
      // ---- tiny frontier: consecutive top-down levels inside one workgroup (bfs_td_small_kernel)
      if (small_nf > 0 && nf <= small_nf && (unsigned long long)scout_count <= small_scout) {
        hipLaunchKernelGGL(bfs_td_small_kernel, dim3(1), dim3(BFS_SMALL_THREADS), 0, 0, g->rowptr, g->colidx, p.visited.p, d_dist,
                           p.q0.p, p.q1.p, qin == p.q1.p ? 1u : 0u, nf, p.qcap, level, (unsigned long long)scout_count, small_nf,
                           small_scout, 1u << 30, p.small_out.p);
        BfsSmallOut so;
        GDN_TRY(p.mail.read(p.small_out.p, so));
        if (so.overflow) {
          gdn_set_error("gdn_bfs: device worklist overflow");
          return GDN_ERR_OVERFLOW;
        }
        iter += (int)so.levels;
        level += (int32_t)so.levels;
        edges_to_check -= (int64_t)so.checked;
        visited_total += (int64_t)so.discovered;
        nf = so.nf;
        scout_count = (int64_t)so.scout;
        qin = so.which ? p.q1.p : p.q0.p;
        qout = so.which ? p.q0.p : p.q1.p;
        light_streak += so.levels;
        lap("small", nf, scout_count);
        continue;
      }
      // ---- light levels beyond one workgroup on a high-diameter graph: the cooperative grid, a barrier per level
      if (coop_nf > 0 && light_streak >= coop_streak && nf <= coop_nf && (unsigned long long)scout_count <= coop_scout) {
        GDN_HIP(hipMemsetAsync(p.coop_cnt.p, 0, 3 * sizeof(BfsCoopCnt), 0));
        GDN_HIP(hipMemsetAsync(p.coop_bar.p, 0, GDN_GBAR_WORDS * sizeof(unsigned), 0));
        const eoff_t *a_rowptr = g->rowptr;
        const vid_t *a_colidx = g->colidx;
        unsigned *a_visited = p.visited.p;
        int32_t *a_depth = d_dist;
        vid_t *a_q0 = p.q0.p, *a_q1 = p.q1.p;
        unsigned a_which = qin == p.q1.p ? 1u : 0u, a_nf = nf, a_cap = p.qcap;
        int32_t a_level = level;
        unsigned long long a_scout = (unsigned long long)scout_count, a_max_scout = coop_scout;
        unsigned a_max_nf = coop_nf, a_min_nf = small_nf ? small_nf / 8u : 0u;
        BfsCoopCnt *a_cnt = p.coop_cnt.p;
        unsigned *a_bar = p.coop_bar.p;
        BfsSmallOut *a_out = p.small_out.p;
        void *args[] = {&a_rowptr, &a_colidx, &a_visited, &a_depth, &a_q0, &a_q1, &a_which, &a_nf, &a_cap, &a_level, &a_scout,
                        &a_max_nf, &a_max_scout, &a_min_nf, &a_cnt, &a_bar, &a_out};
        GDN_HIP(hipLaunchCooperativeKernel((const void *)bfs_td_coop_kernel, dim3((unsigned)p.coop_blocks), dim3(BFS_COOP_THREADS), args,
                                           0, 0));
        BfsSmallOut so;
        GDN_TRY(p.mail.read(p.small_out.p, so));
        if (so.overflow) {
          gdn_set_error("gdn_bfs: device worklist overflow");
          return GDN_ERR_OVERFLOW;
        }
        iter += (int)so.levels;
        level += (int32_t)so.levels;
        edges_to_check -= (int64_t)so.checked;
        visited_total += (int64_t)so.discovered;
        nf = so.nf;
        scout_count = (int64_t)so.scout;
        qin = so.which ? p.q1.p : p.q0.p;
        qout = so.which ? p.q0.p : p.q1.p;
        light_streak += so.levels;
        lap("coop", nf, scout_count);
        continue;
      }
      // ---- top-down step (omp_beamer.cc:143-146)
      ++iter;
      edges_to_check -= scout_count;
      GDN_TRY(cnt_reset());
      BfsTdVis vis;
      vis.rowptr = g->rowptr;
      vis.rec = p.head.p;  // nullptr without heads
      vis.colidx = g->colidx;
      vis.visited = p.visited.p;
      vis.depth = d_dist;
      vis.outq = qout;
      vis.cnt = p.cnt.p;
      vis.cap = p.qcap;
      vis.next_level = level + 1;
      vis.scout_local = 0;
      {  // GDN_BFS_TD_BLIND (test hook): 0 never, 1 always
        const char *eb = gdn_test_option("GDN_BFS_TD_BLIND");
        vis.blind = eb ? eb[0] == '1' : (!p.skewed && p.head.p != nullptr && visited_total * 8 < (int64_t)p.active_rows);
      }
      big.count = &p.cnt.p->big_count;
      big.overflow = &p.cnt.p->overflow;
      // a small frontier of long rows: hand every row of a wave's width or more to the persistent item kernel
      // (rows of 64..511 edges walked one after the other by the few waves of such a level cost 0.35 ms on RMAT-27)
      big.min_deg = ((uint64_t)nf < 65536u && (uint64_t)nf + (uint64_t)scout_count / EXP_CHUNK + 1024u < (uint64_t)p.bigcap)
                        ? 64u : (unsigned)EXP_BIG;
      bool td_keep = false;
      if (p.snap.p && scout_count >= snap_min_edges) {  // the next level may be a bitmap one: see bfs_bitmap_diff_kernel
        GDN_HIP(hipMemcpyAsync(p.snap.p, p.visited.p, (size_t)p.nwords_pad * 4, hipMemcpyDeviceToDevice, 0));
        snap_level = level + 1;
        // deferred depths for a top-down level of some weight too (one scattered 4-byte write per discovery less): its bitmap is
        // the difference to the snapshot, made right behind the level whatever the next one is
        td_keep = defer && maps.n < BFS_DEFER_MAX && scout_count >= td_defer_min;
        if (td_keep) vis.depth = nullptr;
      }
      hipLaunchKernelGGL(bfs_td_kernel, dim3(gdn_nblocks(nf)), dim3(GDN_BLOCK), 0, 0, g->rowptr, qin, nf, big, vis);
      hipLaunchKernelGGL(bfs_td_big_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
      if (td_keep) {
        kept_front = p.lvl.p + (size_t)maps.n * p.nwords_pad;
        kept_front_level = level + 1;
        hipLaunchKernelGGL(bfs_bitmap_diff_kernel, dim3(gdn_nblocks(p.nwords_pad / 4)), dim3(GDN_BLOCK), 0, 0,
                           reinterpret_cast<const uint4 *>(p.visited.p), reinterpret_cast<const uint4 *>(p.snap.p),
                           reinterpret_cast<uint4 *>(kept_front), p.nwords_pad / 4);
        maps.bits[maps.n] = kept_front;
        maps.level[maps.n] = level + 1;
        maps.n++;
      }
      GDN_TRY(cnt_read(h));
      nf = h.next_count;
      scout_count = (int64_t)h.scout;
      visited_total += (int64_t)nf;
      vid_t *t = qin;
      qin = qout;
      qout = t;
      level++;
      light_streak = (coop_nf > 0 && nf <= coop_nf && (unsigned long long)scout_count <= coop_scout) ? light_streak + 1 : 0;
      lap("top-down", nf, scout_count);
    }
    if (h.overflow) {
      gdn_set_error("gdn_bfs: device worklist overflow");
      return GDN_ERR_OVERFLOW;
    }
  }
  if (defer) {  // the distances of the kept levels and of the vertices never reached, in one sequential pass
    if (time_init) GDN_HIP(hipEventRecord(ev.c, 0));
    unsigned fin_blocks = 8192;
    if (const char *e = gdn_xoption("GDN_BFS_FINISH_BLOCKS")) fin_blocks = atoi(e) > 0 ? (unsigned)atoi(e) : fin_blocks;  // (tuning knob)
    const unsigned need_blocks = gdn_nblocks(((uint64_t)m + 3) / 4);
    const dim3 fgrid(need_blocks < fin_blocks ? need_blocks : fin_blocks);
#define BFS_FINISH_LAUNCH(N)                                                                                               \
  case N:                                                                                                                  \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(bfs_depth_finish_kernel<N>), fgrid, dim3(GDN_BLOCK), 0, 0, p.visited.p, maps, d_dist, m, \
                       (int32_t)GDN_MYINFINITY);                                                                           \
    break;
    static_assert(BFS_DEFER_MAX == 8, "one instance of the closing pass per number of kept levels");
    switch (maps.n) {
      BFS_FINISH_LAUNCH(0)
      BFS_FINISH_LAUNCH(1)
      BFS_FINISH_LAUNCH(2)
      BFS_FINISH_LAUNCH(3)
      BFS_FINISH_LAUNCH(4)
      BFS_FINISH_LAUNCH(5)
      BFS_FINISH_LAUNCH(6)
      BFS_FINISH_LAUNCH(7)
      default:
        hipLaunchKernelGGL(HIP_KERNEL_NAME(bfs_depth_finish_kernel<8>), fgrid, dim3(GDN_BLOCK), 0, 0, p.visited.p, maps, d_dist, m,
                           (int32_t)GDN_MYINFINITY);
    }
#undef BFS_FINISH_LAUNCH
    if (time_init) GDN_HIP(hipEventRecord(ev.d, 0));
    GDN_HIP(hipStreamSynchronize(0));
    if (trace) fprintf(stderr, "[bfs] distances of %d kept levels + unreached written at the end  %.3f ms\n", maps.n, tl.stop_ms());
  }
  GDN_HIP(hipGetLastError());
  st.solve_ms = tsolve.stop_ms();
  st.iterations = iter;
  if (time_init) {
    float init_ms = 0.f;
    GDN_HIP(hipEventElapsedTime(&init_ms, ev_a, ev_b));
    st.prep_ms = init_ms;  // fill / bitmap clear / seed: what every BFSSolver of the reference does in FRONT of its Timer
    // The closing pass is reported apart (gdn_stats::last_error, ms) and is NOT initialisation: besides the "unreached" fill it
    // writes the real depths of the kept heavy levels, which the reference writes inside its Timer (BUStep / TDStep,
    // omp_beamer.cc:13-58; the loop at omp_beamer.cc:165-169 behind t.Stop() only maps negative depths to MYINFINITY).
    st.last_error = 0.0;
    if (defer) {
      GDN_HIP(hipEventElapsedTime(&init_ms, ev.c, ev.d));
      st.last_error = init_ms;
    }
  }
  uint64_t te = 0;
  GDN_TRY(gdn_reached_edges(g, d_dist, GDN_MYINFINITY, &te));
  st.edges_traversed = te;
  if (stats) *stats = st;
  return GDN_OK;
}

extern "C" {

int gdn_bfs_plan_create(const gdn_graph *g, const gdn_graph *gin, int32_t dense, gdn_bfs_plan **plan) {
  GDN_REQUIRE(plan != nullptr, "plan");
  *plan = nullptr;
  GDN_REQUIRE(g != nullptr, "graph");
  GDN_REQUIRE(gin == nullptr || gin->m == g->m, "in-CSR vertex count");
  gdn_bfs_plan *p = new gdn_bfs_plan();
  const int rc = bfs_plan_init(*p, g, gin, dense != 0);
  if (rc != GDN_OK) {
    delete p;
    return rc;
  }
  *plan = p;
  return GDN_OK;
}

int gdn_bfs_plan_free(gdn_bfs_plan *plan) {
  delete plan;
  return GDN_OK;
}

int gdn_bfs_run(gdn_bfs_plan *plan, int32_t source, int32_t *d_dist, gdn_stats *stats) {
  GDN_REQUIRE(plan != nullptr && d_dist != nullptr, "plan / d_dist");
  GDN_REQUIRE(source >= 0 && source < plan->g->m, "source out of range");
  return bfs_run(*plan, source, d_dist, stats);
}

int gdn_bfs_dev(const gdn_graph *g, const gdn_graph *gin, int32_t source, int32_t *d_dist, gdn_stats *stats) {
  GDN_REQUIRE(g != nullptr && d_dist != nullptr, "graph / d_dist");
  GDN_REQUIRE(source >= 0 && source < g->m, "source out of range");
  GDN_REQUIRE(gin == nullptr || gin->m == g->m, "in-CSR vertex count");
  gdn_bfs_plan p;
  GDN_TRY(bfs_plan_init(p, g, gin, /*dense=*/false));
  return bfs_run(p, source, d_dist, stats);
}

// Host API: one call == BFSSolver(g, source, dist) (src/bfs/main.cc:22).
int gdn_bfs(int32_t m, uint64_t nnz, const uint64_t *out_rowptr, const int32_t *out_colidx,
            const uint64_t *in_rowptr, const int32_t *in_colidx, int32_t source, int32_t *dist,
            gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && out_rowptr && dist, "null argument");
  GDN_REQUIRE(source >= 0 && source < m, "source out of range");
  GDN_TRY(gdn_require_device());
  HostTimer th2d;
  th2d.start();
  gdn_graph *g = nullptr, *gi = nullptr;
  GDN_TRY(gdn_graph_upload(m, nnz, out_rowptr, out_colidx, &g));
  int rc = GDN_OK;
  DevBuf<int32_t> d_dist;
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  do {
    if (in_rowptr && in_colidx) {
      if (in_rowptr == out_rowptr && in_colidx == out_colidx) gi = g;  // symmetrized graph (csr_graph.h:241-245)
      else if ((rc = gdn_graph_upload(m, nnz, in_rowptr, in_colidx, &gi))) break;
    }
    if ((rc = d_dist.alloc(m))) break;
    const double h2d = th2d.stop_ms();
    if ((rc = gdn_bfs_dev(g, gi, source, d_dist.p, &st))) break;
    st.h2d_ms = h2d;
    if (hipMemcpy(dist, d_dist.p, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_bfs: download failed");
      rc = GDN_ERR_HIP;
    }
  } while (0);
  if (gi && gi != g) gdn_graph_free(gi);
  gdn_graph_free(g);
  if (stats) *stats = st;
  return rc;
}

}  // extern "C"
