// gdn_sssp.hip -- single-source shortest paths: near/far worklists (delta-stepping).
//
// Reference path: SSSPSolver (src/sssp/sssp.h:47).  The OpenMP solver is delta-stepping
// (src/sssp/omp_base.cc:12-97: bins of width delta, relax with a CAS-min :44-55); the live
// CUDA solvers are worklist Bellman-Ford (src/sssp/linear_base.cu:28 bellman_ford with
// atomicMin :43 and one global atomicAdd per pushed vertex; linear_lb.cu:118 the 3-tier lb
// expand) whose queues silently drop on overflow (include/worklistc.h:46-47).  Here:
//   * bucket [lo,hi) of width delta is processed to a fixpoint from the NEAR list,
//     improvements >= hi are parked in the FAR list (the legacy src/sssp/dstep.cu idea);
//   * relax = device-scope atomicMin on dist; a vertex is pushed to NEAR at most once per pass
//     (stamp array) and sits in FAR at most once (flag array), so both lists are bounded by m
//     and never overflow silently;
//   * neighbour expansion = gdn_expand.hpp (hub rows chunked across the grid).
// Distances are exact (integer min is order independent): identical to Dijkstra
// (src/sssp/verifier.cc:8-39).
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gdn_expand.hpp"
#include "gdn_pb.hpp"

// Every hot counter sits on a 128-byte line of its own: atomics on ONE line serialise at ~12-25 ns each whatever issues
// them (a 228 K-vertex pass -- 3.5 K waves, three counters -- spent 0.2 of its 0.26 ms there); lines of their own go
// through different L2 channels side by side.
struct SsspCounters {
  alignas(128) unsigned near_count;
  alignas(128) unsigned far_count;
  alignas(128) unsigned big_count;
  alignas(128) unsigned long long relaxed;
  alignas(128) int max_dist;  // largest distance written since the counters were reset (the dense sweeps size their candidates by it)
  alignas(128) int min_far;
  unsigned overflow;
  alignas(128) unsigned long long improved_edges;  // binned relax pass: out-edges of the rows it improved
};

// record tiers of the dense sweeps (sssp_build_tiers below)
#define SSSP_MAX_TIERS 4
#define SSSP_TIER_ROW_BITS 15               // a record keeps the row of its bin in 15 bits (bins of up to 2^15 rows, as without tiers:
                                            // with 14 bits and 2^14-row bins RMAT-25 / 26 lost 1-3 % where RMAT-24 gained 8 %)
#define SSSP_TIER0 (1u << 15)               // sources of the first tier (a 128 KB table)
#define SSSP_TIER_N (1u << 17)              // of every further one (32 - 15 index bits: 512 KB tables)
struct SsspTierArgs {  // what phase B needs (by value)
  int n = 0;
  unsigned nbins = 0;
  const uint32_t *rec = nullptr;
  const uint8_t *w = nullptr;  // nullptr: every weight is w_uniform
  const eoff_t *ptr = nullptr;
  const uint32_t *cnt = nullptr;  // nullable: records per (tier, bin) stream -- set when the whole 256-record blocks of every stream
                                  // (and of w) are lane-interleaved (PbOutTiers::interleaved); nullptr: plain streams, count = ptr difference
  const unsigned *tab = nullptr;
  unsigned off[SSSP_MAX_TIERS] = {};
  unsigned w_uniform = 0;
};

struct SsspVis {
  const eoff_t *__restrict__ rowptr;
  unsigned long long near_edges;  // per-lane: out-degree sum of the vertices this lane pushed to NEAR
  const vid_t *__restrict__ colidx;
  const int32_t *__restrict__ weight;
  int32_t *__restrict__ dist;
  int32_t *__restrict__ stamp;
  unsigned *__restrict__ in_far;
  vid_t *__restrict__ near_out;
  vid_t *__restrict__ far_out;
  SsspCounters *cnt;
  unsigned cap;
  int32_t thr_hi;
  int32_t pass;
  int32_t no_push;  // != 0: only the distances are lowered, no list is built (the pass in front of the dense sweeps)
  int32_t du;  // per-lane: distance of this lane's source vertex
  int32_t max_d;  // per-lane: largest distance this lane wrote
  GdnWlStage near_st, far_st;  // per-wave LDS strips: one atomic on the hot counters per flush, not per wave step
  __device__ __forceinline__ void begin_big(vid_t v) { du = dist[v]; }
  __device__ __forceinline__ void edge(int owner, eoff_t k, bool valid) {
    const int32_t d_src = __shfl(du, owner, 64);
    bool push_near = false, push_far = false;
    vid_t dst = 0;
    if (valid) {
      dst = __builtin_nontemporal_load(colidx + k);
      const int32_t nd = d_src + __builtin_nontemporal_load(weight + k);
      if (nd < dist[dst]) {
        const int32_t old = atomicMin(&dist[dst], nd);
        if (nd < old) {
          max_d = nd > max_d ? nd : max_d;
          if (no_push) {
            near_edges += 1;  // improved vertices (only a count is kept)
          } else if (nd < thr_hi) {
            push_near = atomicExch(&stamp[dst], pass) != pass;
            if (push_near) near_edges += rowptr[dst + 1] - rowptr[dst];
          } else {
            push_far = atomicExch(&in_far[dst], 1u) == 0u;
          }
        }
      }
    }
    if (no_push) return;  // uniform
    gdn_wl_push_staged(near_st, near_out, &cnt->near_count, cap, push_near, dst, &cnt->overflow);
    gdn_wl_push_staged(far_st, far_out, &cnt->far_count, cap, push_far, dst, &cnt->overflow);
  }
  // by the whole workgroup at the end of the kernel: one reservation per list and one add per counter and WORKGROUP (per
  // wave, a pass over 1.2 M rows put 19 K atomics on each of three addresses: 0.28 ms of its 0.34)
  __device__ __forceinline__ void finish(unsigned *s_tmp, unsigned long long *s_tmp64) {
    gdn_wl_flush_block(near_st, near_out, &cnt->near_count, cap, &cnt->overflow, s_tmp);
    gdn_wl_flush_block(far_st, far_out, &cnt->far_count, cap, &cnt->overflow, s_tmp);
    gdn_block_add_u64(near_edges, &cnt->relaxed, s_tmp64);
    int32_t mx = max_d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int32_t t = __shfl_xor(mx, o, 64);
      mx = t > mx ? t : mx;
    }
    // one hot address: only a wave that raises the maximum touches it
    if (gdn_lane() == 0 && mx > __hip_atomic_load(&cnt->max_dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&cnt->max_dist, mx);
  }
};

#define SSSP_RELAX_GRID 2048u  // 8 workgroups per CU: every wave slot taken, and at most 2048 closing reservations per list
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_relax_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ near_in, unsigned n,
                  int32_t thr_lo, ExpBigList big, SsspVis vis,
                  // nullable: the out-degree sum of the rows this pass expands is added here (the list came from a conversion
                  // that did not sum it: the pass reads the row offsets anyway, the conversion paid two gathers per row)
                  unsigned long long *list_edges = nullptr) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  __shared__ vid_t s_stage[2][GDN_WAVES_PER_BLOCK][GDN_WL_STAGE];
  vis.near_st.strip = s_stage[0][threadIdx.x >> 6];
  vis.far_st.strip = s_stage[1][threadIdx.x >> 6];
  vis.near_st.n = vis.far_st.n = 0;
  vis.max_d = 0;
  __shared__ unsigned s_tmp[GDN_WAVES_PER_BLOCK + 1];
  __shared__ unsigned long long s_tmp64[GDN_WAVES_PER_BLOCK];
  vis.near_edges = 0;
  unsigned long long in_edges = 0;
  // persistent grid (<= SSSP_RELAX_GRID workgroups): the strips fill across the batches of a workgroup
  for (unsigned i0 = blockIdx.x * GDN_BLOCK; i0 < n; i0 += gridDim.x * GDN_BLOCK) {  // block-uniform trip count
    const unsigned i = i0 + threadIdx.x;
    eoff_t b = 0, e = 0;
    vid_t v = 0;
    vis.du = 0;
    if (i < n) {
      v = near_in[i];
      vis.du = vis.dist[v];
      // omp_base.cc:40: entries whose distance fell below the bucket were settled earlier
      if (vis.du >= thr_lo) {
        b = rowptr[v];
        e = rowptr[v + 1];
        in_edges += e - b;
      }
    }
    gdn_expand_wave(b, e, v, big, vis, s_scan[threadIdx.x >> 6]);
  }
  vis.finish(s_tmp, s_tmp64);
  if (list_edges) {  // (uniform) one add per workgroup
    __syncthreads();
    gdn_block_add_u64(in_edges, list_edges, s_tmp64);
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
sssp_relax_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, SsspVis vis) {
  __shared__ vid_t s_stage[2][GDN_WAVES_PER_BLOCK][GDN_WL_STAGE];
  vis.near_st.strip = s_stage[0][threadIdx.x >> 6];
  vis.far_st.strip = s_stage[1][threadIdx.x >> 6];
  vis.near_st.n = vis.far_st.n = 0;
  vis.du = 0;
  vis.max_d = 0;
  vis.near_edges = 0;
  __shared__ unsigned s_tmp[GDN_WAVES_PER_BLOCK + 1];
  __shared__ unsigned long long s_tmp64[GDN_WAVES_PER_BLOCK];
  gdn_expand_big_items(rowptr, big, vis);
  vis.finish(s_tmp, s_tmp64);
}

// FAR -> {NEAR of the new bucket, FAR kept (and the smallest distance kept: the jump over empty buckets), dropped}
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_far_split_kernel(const vid_t *__restrict__ far_in, unsigned n, const int32_t *__restrict__ dist,
                      int32_t old_hi, int32_t new_hi, unsigned *__restrict__ in_far, vid_t *__restrict__ near_out,
                      vid_t *__restrict__ far_out, SsspCounters *cnt, unsigned cap, const eoff_t *__restrict__ rowptr) {
  // persistent grid, staged pushes: one atomic per list and ~4 wave steps instead of two per wave step (a far list of
  // millions of entries made the two hot counters the cost of the split: 0.44 ms per bucket on RMAT-24)
  __shared__ vid_t s_near[GDN_WAVES_PER_BLOCK][GDN_WL_STAGE], s_far[GDN_WAVES_PER_BLOCK][GDN_WL_STAGE];
  GdnWlStage st_near, st_far;
  st_near.strip = s_near[threadIdx.x >> 6];
  st_near.n = 0;
  st_far.strip = s_far[threadIdx.x >> 6];
  st_far.n = 0;
  unsigned long long deg = 0;
  int32_t dmin = GDN_DIST_INF;  // smallest distance of what stays in FAR (the host's jump over empty buckets)
  const unsigned stride = gridDim.x * GDN_BLOCK;
  for (unsigned i0 = blockIdx.x * GDN_BLOCK; i0 < n; i0 += stride) {  // wave-uniform trip count
    const unsigned i = i0 + threadIdx.x;
    bool to_near = false, to_far = false;
    vid_t w = 0;
    if (i < n) {
      w = far_in[i];
      const int32_t d = dist[w];
      if (d >= new_hi) {
        to_far = true;
        dmin = d < dmin ? d : dmin;
      } else {
        in_far[w] = 0u;
        to_near = d >= old_hi;
        if (to_near) deg += rowptr[w + 1] - rowptr[w];
      }
    }
    gdn_wl_push_staged(st_near, near_out, &cnt->near_count, cap, to_near, w, &cnt->overflow);
    gdn_wl_push_staged(st_far, far_out, &cnt->far_count, cap, to_far, w, &cnt->overflow);
  }
  gdn_wl_flush(st_near, near_out, &cnt->near_count, cap, &cnt->overflow);
  gdn_wl_flush(st_far, far_out, &cnt->far_count, cap, &cnt->overflow);
  deg = gdn_wave_sum(deg);
  if (gdn_lane() == 0 && deg) atomicAdd(&cnt->relaxed, deg);
  // one atomicMin per WORKGROUP (a single hot address)
  __shared__ int32_t s_dmin[GDN_WAVES_PER_BLOCK];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int32_t t = __shfl_xor(dmin, o, 64);
    dmin = t < dmin ? t : dmin;
  }
  if (gdn_lane() == 0) s_dmin[threadIdx.x >> 6] = dmin;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < GDN_WAVES_PER_BLOCK; w++) dmin = s_dmin[w] < dmin ? s_dmin[w] : dmin;
    if (dmin != GDN_DIST_INF) atomicMin(&cnt->min_far, dmin);
  }
}

__global__ void sssp_seed_kernel(int32_t source, int32_t *dist, vid_t *near) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    dist[source] = 0;
    near[0] = source;
  }
}

int gdn_reached_edges(const gdn_graph *g, const int32_t *d_dist, int32_t unreached, uint64_t *out);

// ------------------------------------------------------------------------------------------
// Dense relaxation sweep = one Bellman-Ford pass over ALL edges on the propagation-blocked layout
// (gdn_pb.hpp, built from the out-CSR with the weights permuted into tile order).  It replaces the
// worklist passes while the frontier is heavy: those are one divergent probe + one global atomicMin
// per edge (7 GTEPS on RMAT-24), a sweep streams ~16 B per edge.
//   phase A (per source chunk): dist[chunk] -> LDS; candidate = dist[u] + w for every edge, stored at
//           the edge's bin-major place (INF stays INF)
//   phase B (per destination bin): ds_min_u32 into the bin's LDS minima, then dist[v] = min(dist[v], .)
//           for the bin's rows; improved rows are counted and flagged in a bitmap (the next worklist)
// ------------------------------------------------------------------------------------------
// Bytes per edge are what a sweep costs, so both per-edge streams are as narrow as the DATA allows (decided per plan /
// per sweep on the host, every variant exact):
//   weights    WB = 0: all weights equal (no stream at all -- unit weights, the reference main's input, src/sssp/main.cc:26),
//              1 / 2: every weight < 2^8 / 2^16 (narrowed once at plan build), 4: int32 as given
//   candidates CT = u8 / u16 while (largest finite distance + largest weight) < 0xFF / 0xFFFF -- the host tracks the largest
//              distance written (SsspCounters::max_dist) --, u32 otherwise; INF is the all-ones pattern of the type
// One thread handles one GROUP of 8 edges per step (16 B of U, 8 x WB B of weights, one G entry, 8 x sizeof(CT) B out).
typedef unsigned sssp_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sssp_u32x2 __attribute__((ext_vector_type(2)));

template <typename CT>
struct SsspCand;
template <>
struct SsspCand<uint16_t> {
  static constexpr unsigned INF = 0xFFFFu;
};
template <>
struct SsspCand<uint32_t> {
  static constexpr unsigned INF = 0xFFFFFFFFu;
};
template <>
struct SsspCand<uint8_t> {
  static constexpr unsigned INF = 0xFFu;
};

template <int WB, typename CT>
__global__ void __launch_bounds__(PB_THREADS)
sssp_pb_expand_kernel(const int32_t *__restrict__ dist, int32_t m_src, int log_chunk,
                      const eoff_t *__restrict__ chunk_ptr, const uint32_t *__restrict__ chunk_order,
                      const uint16_t *__restrict__ U, const uint32_t *__restrict__ G,
                      const void *__restrict__ Wv, unsigned w_uniform, CT *__restrict__ cand, unsigned *__restrict__ bad) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_d[];
  const unsigned ch = 1u << log_chunk;
  const unsigned c = chunk_order[blockIdx.x];
  const size_t base = (size_t)c << log_chunk;
  for (unsigned i = threadIdx.x; i < ch; i += PB_THREADS) {
    const size_t g = base + i;
    s_d[i] = (g < (size_t)m_src) ? (unsigned)dist[g] : (unsigned)GDN_DIST_INF;
  }
  if (threadIdx.x == 0) s_d[ch] = (unsigned)GDN_DIST_INF;  // pad edges
  __syncthreads();
  const eoff_t g0 = chunk_ptr[c] >> 3, g1 = chunk_ptr[c + 1] >> 3;
  const sssp_u32x4 *U8 = reinterpret_cast<const sssp_u32x4 *>(U);
  constexpr int UNR = 4;
  constexpr unsigned CINF = SsspCand<CT>::INF;
  unsigned overflowed = 0;
  for (eoff_t g = g0 + threadIdx.x; g < g1; g += UNR * PB_THREADS) {
    sssp_u32x4 u[UNR];
    unsigned w[UNR][8];
    unsigned d[UNR];
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t gg = g + (eoff_t)r * PB_THREADS;
      if (gg < g1) {
        u[r] = __builtin_nontemporal_load(U8 + gg);
        d[r] = __builtin_nontemporal_load(G + gg);
        if (WB == 1) {
          const sssp_u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const sssp_u32x2 *>(Wv) + gg);
#pragma unroll
          for (int k = 0; k < 8; k++) w[r][k] = ((k < 4 ? t.x : t.y) >> (8 * (k & 3))) & 0xFFu;
        } else if (WB == 2) {
          const sssp_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const sssp_u32x4 *>(Wv) + gg);
#pragma unroll
          for (int k = 0; k < 8; k++) w[r][k] = (t[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        } else if (WB == 4) {
          const sssp_u32x4 t0 = __builtin_nontemporal_load(reinterpret_cast<const sssp_u32x4 *>(Wv) + 2 * gg);
          const sssp_u32x4 t1 = __builtin_nontemporal_load(reinterpret_cast<const sssp_u32x4 *>(Wv) + 2 * gg + 1);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            w[r][k] = t0[k];
            w[r][4 + k] = t1[k];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; k++) w[r][k] = w_uniform;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t gg = g + (eoff_t)r * PB_THREADS;
      if (gg < g1) {
        unsigned o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const unsigned uu = (u[r][k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
          const unsigned t = s_d[uu];
          unsigned nd = t + w[r][k];  // < 2^32: both are <= INT_MAX
          if (t >= (unsigned)GDN_DIST_INF || nd >= (unsigned)GDN_DIST_INF) nd = CINF;  // no path yet (or beyond the int range)
          else if (sizeof(CT) < 4 && nd >= CINF) {
            overflowed = 1u;  // cannot happen: the host switches to 32-bit candidates before a distance gets here
            nd = CINF;
          }
          o[k] = nd;
        }
        if (sizeof(CT) == 1) {
          sssp_u32x2 v;
          v.x = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
          v.y = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
          reinterpret_cast<sssp_u32x2 *>(cand)[(size_t)d[r]] = v;
        } else if (sizeof(CT) == 2) {
          sssp_u32x4 v;
#pragma unroll
          for (int k = 0; k < 4; k++) v[k] = o[2 * k] | (o[2 * k + 1] << 16);
          reinterpret_cast<sssp_u32x4 *>(cand)[(size_t)d[r]] = v;
        } else {
          sssp_u32x4 v0, v1;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            v0[k] = o[k];
            v1[k] = o[4 + k];
          }
          reinterpret_cast<sssp_u32x4 *>(cand)[2 * (size_t)d[r]] = v0;
          reinterpret_cast<sssp_u32x4 *>(cand)[2 * (size_t)d[r] + 1] = v1;
        }
      }
    }
  }
  if (overflowed) *bad = 1u;
}

template <typename CT>
__global__ void __launch_bounds__(PB_THREADS)
sssp_pb_accumulate_kernel(int32_t m_dst, int log_bin, const eoff_t *__restrict__ bin_ptr,
                          const uint32_t *__restrict__ bin_order, const uint16_t *__restrict__ V,
                          const CT *__restrict__ cand, int32_t *__restrict__ dist,
                          unsigned *__restrict__ improved_bits, SsspCounters *cnt,
                          const eoff_t *__restrict__ out_rowptr,  // nullable: count the improved rows' out-edges
                          const SsspTierArgs ta) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_min[];
  __shared__ unsigned long long s_red[2 * PB_WAVES];
  __shared__ int s_max[PB_WAVES];
  const unsigned bn = 1u << log_bin;
  const unsigned b = bin_order[blockIdx.x];
  // (measured and dropped: the fold started from the rows' current distances with an LDS read in front of every atomic -- most
  // candidates lose from the second sweep on -- 383 -> 400 us)
  for (unsigned i = threadIdx.x; i < bn; i += PB_THREADS) s_min[i] = (unsigned)GDN_DIST_INF;
  __syncthreads();
  const eoff_t q0 = bin_ptr[b] >> 3, q1 = bin_ptr[b + 1] >> 3;
  const sssp_u32x4 *V8 = reinterpret_cast<const sssp_u32x4 *>(V);
  const sssp_u32x4 *C = reinterpret_cast<const sssp_u32x4 *>(cand);
  constexpr int UNR = 4;
  constexpr unsigned CINF = SsspCand<CT>::INF;
  for (eoff_t q = q0 + threadIdx.x; q < q1; q += UNR * PB_THREADS) {
    sssp_u32x4 xs[UNR][sizeof(CT) == 4 ? 2 : 1];
    sssp_u32x4 vs[UNR];
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t qq = q + (eoff_t)r * PB_THREADS;
      if (qq < q1) {
        if (sizeof(CT) == 1) {
          const sssp_u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const sssp_u32x2 *>(cand) + qq);
          xs[r][0].x = t.x;
          xs[r][0].y = t.y;
        } else if (sizeof(CT) == 2) xs[r][0] = __builtin_nontemporal_load(C + qq);
        else {
          xs[r][0] = __builtin_nontemporal_load(C + 2 * qq);
          xs[r][sizeof(CT) == 4 ? 1 : 0] = __builtin_nontemporal_load(C + 2 * qq + 1);
        }
        vs[r] = __builtin_nontemporal_load(V8 + qq);
      }
    }
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t qq = q + (eoff_t)r * PB_THREADS;
      if (qq < q1) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const unsigned x = sizeof(CT) == 1   ? ((xs[r][0][k >> 2] >> (8 * (k & 3))) & 0xFFu)
                             : sizeof(CT) == 2 ? ((xs[r][0][k >> 1] >> (16 * (k & 1))) & 0xFFFFu)
                                               : xs[r][sizeof(CT) == 4 ? (k >> 2) : 0][k & 3];
          const unsigned v = (vs[r][k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
          if (x != CINF) atomicMin(&s_min[v], x);
        }
      }
    }
  }
  // the record tiers: ONE record per lane and load, eight loads in flight (records are sorted by source inside a bin: the 64
  // table reads of a wave instruction then fall into a few consecutive lines -- near-coalesced L2 hits; four consecutive
  // records per lane spread them over four times as many lines and cost the sweep 0.66 instead of 0.6 ms)
  for (int t = 0; t < ta.n; t++) {
    const eoff_t j0 = ta.ptr[(size_t)t * ta.nbins + b];
    const unsigned nr = ta.cnt ? ta.cnt[(size_t)t * ta.nbins + b] : (unsigned)(ta.ptr[(size_t)t * ta.nbins + b + 1] - j0);
    const unsigned *__restrict__ tab = ta.tab + ta.off[t];
    const uint32_t *__restrict__ R = ta.rec + j0;
    const uint8_t *__restrict__ W = ta.w ? ta.w + j0 : nullptr;
    constexpr int TU = 8;
    constexpr unsigned RMASK = (1u << SSSP_TIER_ROW_BITS) - 1u;
    // interleaved streams: the whole blocks of 256 records with ONE 16-byte record load and ONE 4-byte weight load per lane
    // and four records (lane l of a block holds records l, 64 + l, 128 + l, 192 + l of the sorted stream, so table read j of
    // the wave still covers 64 consecutive records); a quarter of the record loads and of the weight loads of the plain form
    const unsigned nfull = ta.cnt ? nr & ~255u : 0u;
    {
      const sssp_u32x4 *__restrict__ R4 = reinterpret_cast<const sssp_u32x4 *>(R);  // j0 is a multiple of 256
      const uint32_t *__restrict__ W4 = reinterpret_cast<const uint32_t *>(W);
      constexpr int IU = 2;
      const unsigned nq = nfull >> 2;
      for (unsigned i0 = threadIdx.x; i0 < nq; i0 += (unsigned)IU * PB_THREADS) {
        sssp_u32x4 rc[IU];
        unsigned wv[IU], d[IU][4];
        bool on[IU];
#pragma unroll
        for (int r = 0; r < IU; r++) {
          const unsigned i = i0 + (unsigned)r * PB_THREADS;
          on[r] = i < nq;
          rc[r] = sssp_u32x4{0u, 0u, 0u, 0u};
          wv[r] = 0u;
          if (on[r]) {
            rc[r] = __builtin_nontemporal_load(R4 + i);
            if (W4) wv[r] = __builtin_nontemporal_load(W4 + i);
          }
        }
#pragma unroll
        for (int r = 0; r < IU; r++)
#pragma unroll
          for (int j = 0; j < 4; j++) d[r][j] = on[r] ? tab[rc[r][j] >> SSSP_TIER_ROW_BITS] : (unsigned)GDN_DIST_INF;
#pragma unroll
        for (int r = 0; r < IU; r++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const unsigned nd = d[r][j] + (W4 ? ((wv[r] >> (8 * j)) & 0xFFu) : ta.w_uniform);
            if (d[r][j] < (unsigned)GDN_DIST_INF && nd < (unsigned)GDN_DIST_INF) atomicMin(&s_min[rc[r][j] & RMASK], nd);
          }
      }
    }
    for (unsigned i0 = nfull + threadIdx.x; i0 < nr; i0 += (unsigned)TU * PB_THREADS) {
      unsigned rc[TU], wv[TU], d[TU];
      bool on[TU];
#pragma unroll
      for (int r = 0; r < TU; r++) {
        const unsigned i = i0 + (unsigned)r * PB_THREADS;
        on[r] = i < nr;
        rc[r] = 0u;
        wv[r] = ta.w_uniform;
        if (on[r]) {
          rc[r] = __builtin_nontemporal_load(R + i);
          if (W) wv[r] = W[i];
        }
      }
#pragma unroll
      for (int r = 0; r < TU; r++) d[r] = on[r] ? tab[rc[r] >> SSSP_TIER_ROW_BITS] : (unsigned)GDN_DIST_INF;
#pragma unroll
      for (int r = 0; r < TU; r++) {
        const unsigned nd = d[r] + wv[r];
        if (d[r] < (unsigned)GDN_DIST_INF && nd < (unsigned)GDN_DIST_INF) atomicMin(&s_min[rc[r] & RMASK], nd);
      }
    }
  }
  __syncthreads();
  // epilogue: one row per thread and step; a wave covers 64 consecutive rows = 2 bitmap words
  const unsigned lane = gdn_lane();
  const size_t row0 = (size_t)b << log_bin;
  unsigned long long improved = 0, edges = 0;
  int mx = 0;
  for (unsigned i = threadIdx.x; i < bn; i += PB_THREADS) {
    const size_t row = row0 + i;
    bool imp = false;
    if (row < (size_t)m_dst) {
      const unsigned nm = s_min[i];
      const unsigned old = (unsigned)dist[row];
      if (nm < old) {
        dist[row] = (int32_t)nm;
        mx = (int)nm > mx ? (int)nm : mx;
        if (out_rowptr) edges += out_rowptr[row + 1] - out_rowptr[row];
        imp = true;
      }
    }
    const unsigned long long mask = __ballot(imp);
    if ((lane & 31u) == 0) improved_bits[(row0 + i) >> 5] = (unsigned)(mask >> (lane & 32u));
    if (lane == 0) improved += (unsigned long long)__popcll(mask);
  }
  improved = gdn_wave_sum(improved);
  edges = gdn_wave_sum(edges);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(mx, o, 64);
    mx = t > mx ? t : mx;
  }
  if (lane == 0) {
    s_red[threadIdx.x >> 6] = improved;
    s_red[PB_WAVES + (threadIdx.x >> 6)] = edges;
    s_max[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0, te = 0;
    int m2 = 0;
    for (int i = 0; i < PB_WAVES; i++) {
      t += s_red[i];
      te += s_red[PB_WAVES + i];
      m2 = s_max[i] > m2 ? s_max[i] : m2;
    }
    if (t) atomicAdd(&cnt->relaxed, t);
    if (te) atomicAdd(&cnt->improved_edges, te);
    if (m2 > __hip_atomic_load(&cnt->max_dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&cnt->max_dist, m2);
  }
}

// weights in tile order, narrowed to WT (plan build; every weight fits by the host's check of the maximum)
template <typename WT>
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_narrow_weights_kernel(const uint32_t *__restrict__ w, size_t n, WT *__restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * GDN_BLOCK) out[i] = (WT)w[i];
}

// [0] = min, [1] = max of the weights (as signed ints), [2] = 1 if any is negative
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_weight_range_kernel(const int32_t *__restrict__ w, size_t n, int32_t *__restrict__ out) {
  int32_t lo = 0x7FFFFFFF, hi = -0x7FFFFFFF - 1;
  for (size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * GDN_BLOCK) {
    const int32_t x = w[i];
    lo = x < lo ? x : lo;
    hi = x > hi ? x : hi;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int32_t a = __shfl_xor(lo, o, 64), b2 = __shfl_xor(hi, o, 64);
    lo = a < lo ? a : lo;
    hi = b2 > hi ? b2 : hi;
  }
  if (gdn_lane() == 0) {
    atomicMin(&out[0], lo);
    atomicMax(&out[1], hi);
  }
}

// ------------------------------------------------------------------------------------------
// Light phases without the host (the reference's "fusion" idea, src/bfs/fusion.cu:163 / include/gbar.h, for the part of a
// search where a grid has nothing to do): ONE 1024-thread workgroup runs consecutive relax passes AND bucket changes --
// the minimum over FAR, the split into the next bucket -- while the NEAR list holds at most SSSP_SMALL_V vertices /
// SSSP_SMALL_E out-edges and the FAR list at most SSSP_SMALL_FAR entries; pass and bucket boundaries are
// __syncthreads().  A pass costs a few microseconds here instead of two launches and a blocking read back.  The lists are
// written and read by this workgroup only, but through different waves and in alternating roles: device-scope accesses.
// ------------------------------------------------------------------------------------------
#define SSSP_SMALL_THREADS 1024
#define SSSP_SMALL_V 512u
#define SSSP_SMALL_E 8192ull
#define SSSP_SMALL_FAR 8192u   // (65536 until round 4: a longer FAR list changes buckets faster on the grid -- one speculative split, RMAT-24 U[1,255] 2.88 -> 2.82 ms)
struct SsspSmallState {
  unsigned n_near, n_far;
  unsigned long long near_edges;
  long long thr_lo, thr_hi;
  int pass;
  int status;  // out: 0 finished (both lists empty), 1 the NEAR list outgrew the workgroup, 2 the FAR list did (next bucket)
  unsigned near_sel, far_sel;  // which buffer of each pair is current
  unsigned overflow;
  int max_dist;
  unsigned passes, buckets;
  unsigned long long relaxed_edges;  // out: out-degree sum of the NEAR lists the passes of this launch walked
  // the bucket width the schedule works with (in / out): starts at the caller's delta and adapts at every bucket change --
  // a bucket whose passes relaxed fewer than `light` edges was all latency, the next one is twice as wide; more than
  // 8 x light: half (never below the caller's).  Distances do not depend on the widths (every bucket runs to its fixpoint).
  long long delta_cur;
  unsigned long long bucket_work;  // edges relaxed in the current bucket so far (in / out)
  unsigned long long light;        // in: 0 = no adaptation (delta_cur stays)
  unsigned light_run, adapt_after; // light buckets in a row so far (in / out); widening starts behind `adapt_after` of them: the
                                   // handful of light buckets in front of an R-MAT search's heavy phases keep the caller's
                                   // width (merged, they hand the dense sweeps a worse start: RMAT-24 2.9 -> 3.2 ms)
};
#define SSSP_DELTA_MAX (1ll << 24)
__device__ __host__ inline long long sssp_adapt_delta(long long cur, long long floor_, unsigned long long work, unsigned long long light,
                                                      unsigned &light_run, unsigned adapt_after) {
  if (light == 0ull) return cur;
  if (work < light) {
    light_run++;
    return (light_run > adapt_after && cur * 2 <= SSSP_DELTA_MAX) ? cur * 2 : cur;
  }
  light_run = 0u;
  if (work > 8ull * light) return cur / 2 >= floor_ ? cur / 2 : floor_;
  return cur;
}

// wave-aggregated slot reservation on an LDS counter (all lanes of the wave must call it; a lane per item serialises on
// the one address: 34 K far entries cost 0.1 ms that way)
__device__ __forceinline__ unsigned sssp_lds_slot(unsigned *counter, bool want) {
  const unsigned long long mask = __ballot(want);
  if (mask == 0ull) return 0u;
  const int leader = __ffsll((long long)mask) - 1;
  unsigned base = 0;
  if ((int)gdn_lane() == leader) base = atomicAdd(counter, (unsigned)__popcll(mask));
  return __shfl(base, leader, 64) + (unsigned)__popcll(mask & gdn_lanemask_lt());
}

__device__ __forceinline__ vid_t sssp_ld(const vid_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sssp_st(vid_t *p, vid_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(SSSP_SMALL_THREADS)
sssp_small_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const int32_t *__restrict__ weight,
                  int32_t *dist, int32_t *stamp, unsigned *in_far, vid_t *near0, vid_t *near1, vid_t *far0, vid_t *far1,
                  unsigned cap, int32_t delta, unsigned max_v, unsigned long long max_e, unsigned max_far,
                  SsspSmallState *state) {
  __shared__ unsigned s_nn, s_nf, s_over;
  __shared__ unsigned long long s_edges;
  __shared__ int s_min, s_maxd;
  const unsigned lane = gdn_lane(), wave = threadIdx.x >> 6, nwaves = SSSP_SMALL_THREADS / 64;
  unsigned n_near = state->n_near, n_far = state->n_far;
  unsigned long long near_edges = state->near_edges;
  long long thr_lo = state->thr_lo, thr_hi = state->thr_hi;
  int pass = state->pass, status = 0;
  unsigned near_sel = state->near_sel, far_sel = state->far_sel, passes = 0, buckets = 0;
  unsigned long long relaxed_edges = 0;
  long long dlt = state->delta_cur;
  unsigned long long bwork = state->bucket_work;
  const unsigned long long light = state->light;
  unsigned light_run = state->light_run;
  const unsigned adapt_after = state->adapt_after;
  auto clamp = [](long long x) { return (int32_t)(x > GDN_DIST_INF ? GDN_DIST_INF : x); };
  if (threadIdx.x == 0) {
    s_over = 0u;
    s_maxd = 0;
  }
  __syncthreads();
  for (;;) {
    if (n_near > 0) {
      // ---- one relax pass over NEAR (sssp_relax_kernel's work)
      ++pass;
      ++passes;
      relaxed_edges += near_edges;
      bwork += near_edges;
      if (threadIdx.x == 0) {
        s_nn = 0u;
        s_nf = n_far;
        s_edges = 0ull;
      }
      __syncthreads();
      const vid_t *near_in = near_sel ? near1 : near0;
      vid_t *near_out = near_sel ? near0 : near1;
      vid_t *far_cur = far_sel ? far1 : far0;
      const int32_t lo = clamp(thr_lo), hi = clamp(thr_hi);
      int32_t maxd = 0;
      // 64 list entries at a time, one per lane (every wave loads the batch: 64 parallel loads instead of a chain of
      // dependent ones); a row of 256 edges or more is walked by ALL waves together, shorter ones by one wave each
      for (unsigned base = 0; base < n_near; base += 64) {
        const unsigned i = base + lane;
        eoff_t bb = 0, ee = 0;
        int32_t dd = 0;
        if (i < n_near) {
          const vid_t v = sssp_ld(near_in + i);
          dd = __hip_atomic_load(dist + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (dd >= lo) {  // else: settled in an earlier bucket (omp_base.cc:40)
            bb = rowptr[v];
            ee = rowptr[v + 1];
          }
        }
        const unsigned cnt = n_near - base < 64u ? n_near - base : 64u;
        for (unsigned j = 0; j < cnt; j++) {
          const eoff_t b = __shfl(bb, (int)j, 64), e = __shfl(ee, (int)j, 64);
          if (e == b) continue;
          const bool coop = e - b >= 256u;
          if (!coop && (j & (nwaves - 1u)) != wave) continue;
          const int32_t du = __shfl(dd, (int)j, 64);
          const unsigned step = coop ? (unsigned)SSSP_SMALL_THREADS : 64u;
          for (eoff_t k = b + (coop ? threadIdx.x : lane); k < e; k += step) {
            const vid_t dst = colidx[k];
            const int32_t nd = du + weight[k];
            if (nd < __hip_atomic_load(dist + dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
              const int32_t old = atomicMin(&dist[dst], nd);
              if (nd < old) {
                maxd = nd > maxd ? nd : maxd;
                if (nd < hi) {
                  if (atomicExch(&stamp[dst], pass) != pass) {
                    const unsigned pos = atomicAdd(&s_nn, 1u);
                    if (pos < cap) sssp_st(near_out + pos, dst);
                    else s_over = 1u;
                    atomicAdd(&s_edges, (unsigned long long)(rowptr[dst + 1] - rowptr[dst]));
                  }
                } else if (atomicExch(&in_far[dst], 1u) == 0u) {
                  const unsigned pos = atomicAdd(&s_nf, 1u);
                  if (pos < cap) sssp_st(far_cur + pos, dst);
                  else s_over = 1u;
                }
              }
            }
          }
        }
      }
      if (maxd > 0) atomicMax(&s_maxd, maxd);
      gdn_wg_level_sync();
      n_near = s_nn;
      n_far = s_nf;
      near_edges = s_edges;
      near_sel ^= 1u;
      __syncthreads();  // everybody has read the counters before they are reset
      if (s_over) break;
      if (n_near > 0) {
        if (n_near > max_v || near_edges > max_e) {
          status = 1;
          break;
        }
        continue;
      }
    }
    // ---- NEAR is empty: the next non-empty bucket (omp_base.cc:66-72)
    if (n_far == 0) {
      status = 0;
      break;
    }
    if (n_far > max_far) {
      status = 2;
      break;
    }
    // One pass per bucket change where the reference's scan needs two (the minimum over FAR, then the split): the split
    // is run at once for the bucket right BEHIND the old one -- on a graph whose buckets are dense (a road-like lattice:
    // tens of thousands of them) that IS the next non-empty bucket -- and collects the minimum of what it keeps on the
    // way; only when nothing moved does a second pass split at the bucket of that minimum (omp_base.cc:66-72 finds the
    // same bucket either way).
    vid_t *near_in = near_sel ? near1 : near0;
    dlt = sssp_adapt_delta(dlt, (long long)delta, bwork, light, light_run, adapt_after);
    bwork = 0ull;
    long long spec_lo = thr_hi, spec_hi = thr_hi + dlt;
    for (;;) {
      vid_t *far_cur = far_sel ? far1 : far0, *far_nxt = far_sel ? far0 : far1;
      if (threadIdx.x == 0) {
        s_min = GDN_DIST_INF;
        s_nn = 0u;
        s_nf = 0u;
        s_edges = 0ull;
      }
      __syncthreads();
      {
        const int32_t ohi = clamp(thr_hi), nhi = clamp(spec_hi);
        unsigned long long deg_sum = 0;
        int32_t dmin = GDN_DIST_INF;
        for (unsigned i0 = wave * 64u; i0 < n_far; i0 += 4 * SSSP_SMALL_THREADS) {  // wave-uniform bounds
          vid_t w[4];
          int32_t d[4];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const unsigned i = i0 + lane + (unsigned)r * SSSP_SMALL_THREADS;
            w[r] = i < n_far ? sssp_ld(far_cur + i) : -1;
          }
#pragma unroll
          for (int r = 0; r < 4; r++) d[r] = w[r] >= 0 ? __hip_atomic_load(dist + w[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
          for (int r = 0; r < 4; r++) {  // (wave-uniform trip count: the reservations below are convergent)
            const bool live = w[r] >= 0;
            const bool to_far = live && d[r] >= nhi;
            const bool to_near = live && !to_far && d[r] >= ohi;  // below the old bucket: a stale entry, dropped
            if (live && !to_far) __hip_atomic_store(in_far + w[r], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned pf = sssp_lds_slot(&s_nf, to_far);
            if (to_far) {
              sssp_st(far_nxt + pf, w[r]);
              dmin = d[r] < dmin ? d[r] : dmin;
            }
            const unsigned pn = sssp_lds_slot(&s_nn, to_near);
            if (to_near) {
              sssp_st(near_in + pn, w[r]);
              deg_sum += rowptr[w[r] + 1] - rowptr[w[r]];
            }
          }
        }
        deg_sum = gdn_wave_sum(deg_sum);
        if (lane == 0 && deg_sum) atomicAdd(&s_edges, deg_sum);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const int32_t t = __shfl_xor(dmin, o, 64);
          dmin = t < dmin ? t : dmin;
        }
        if (lane == 0 && dmin != GDN_DIST_INF) atomicMin(&s_min, dmin);
      }
      gdn_wg_level_sync();
      const unsigned nn = s_nn, nf = s_nf;
      const int32_t mn = s_min;
      near_edges = s_edges;
      far_sel ^= 1u;
      n_far = nf;
      __syncthreads();
      if (nn > 0 || nf == 0) {
        if (nn > 0) {
          thr_lo = spec_lo;
          thr_hi = spec_hi;
          ++buckets;
        }
        n_near = nn;
        break;
      }
      thr_hi = spec_hi;  // that bucket was empty and everything kept lies behind it: nothing is stale against it
      spec_lo = ((long long)mn / dlt) * dlt;  // (the caller's grid of buckets while the width is the caller's)
      if (spec_lo < thr_hi) spec_lo = thr_hi;  // widths that changed on the way: never back into what has been settled
      spec_hi = spec_lo + dlt;
    }
    if (n_near > max_v || near_edges > max_e) {
      status = 1;
      break;
    }
  }
  if (threadIdx.x == 0) {
    state->n_near = n_near;
    state->n_far = n_far;
    state->near_edges = near_edges;
    state->thr_lo = thr_lo;
    state->thr_hi = thr_hi;
    state->pass = pass;
    state->status = status;
    state->near_sel = near_sel;
    state->far_sel = far_sel;
    state->overflow = s_over;
    state->max_dist = s_maxd;
    state->passes = passes;
    state->buckets = buckets;
    state->relaxed_edges = relaxed_edges;
    state->delta_cur = dlt;
    state->bucket_work = bwork;
    state->light_run = light_run;
  }
}

// ------------------------------------------------------------------------------------------
// The same loop -- relax passes AND bucket changes -- on a COOPERATIVE grid (one workgroup per CU, a grid barrier per
// phase) for the lists that outgrow one workgroup on a high-diameter graph: a road-like lattice with U[1,255] weights
// takes tens of thousands of buckets of a few thousand vertices each, 33 us per phase on the host loop.  The CDNA form of
// the reference's persistent kernels over a software global barrier (include/gbar.h:24-65, src/sssp fusion variants).
// Counters: three rotating sets, one per phase (the set of phase p+1 is reset while phase p runs), each on cache lines
// of its own; a wave takes 64 list entries, one per lane: short rows are walked lane-private (64 rows in flight), rows
// of a wave's width or more by the whole wave.
// ------------------------------------------------------------------------------------------
#define SSSP_COOP_THREADS 256
struct SsspCoopCnt {
  alignas(128) unsigned nn;        // NEAR entries produced by the phase
  alignas(128) unsigned nf;        // FAR entries appended (relax) / kept (split)
  alignas(128) unsigned long long edges;
  alignas(128) int min_far;
  unsigned over;
};

// wave-aggregated slot reservation on a GLOBAL counter (all lanes of the wave must call it)
__device__ __forceinline__ unsigned sssp_global_slot(unsigned *counter, bool want) {
  const unsigned long long mask = __ballot(want);
  if (mask == 0ull) return 0u;
  const int leader = __ffsll((long long)mask) - 1;
  unsigned base = 0;
  if ((int)gdn_lane() == leader) base = atomicAdd(counter, (unsigned)__popcll(mask));
  return __shfl(base, leader, 64) + (unsigned)__popcll(mask & gdn_lanemask_lt());
}

__global__ void __launch_bounds__(SSSP_COOP_THREADS)
sssp_coop_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const int32_t *__restrict__ weight,
                 int32_t *dist, int32_t *stamp, unsigned *in_far, vid_t *near0, vid_t *near1, vid_t *far0, vid_t *far1,
                 unsigned cap, int32_t delta, unsigned max_v, unsigned long long max_e, unsigned max_far,
                 SsspCoopCnt *cnt /* 3 sets, reset by the host */, unsigned *bar /* GDN_GBAR_WORDS, zeroed by the host */,
                 SsspSmallState *state) {
  const unsigned lane = gdn_lane();
  const unsigned gt = blockIdx.x * SSSP_COOP_THREADS + threadIdx.x, nt = gridDim.x * SSSP_COOP_THREADS;
  unsigned n_near = state->n_near, n_far = state->n_far;
  unsigned long long near_edges = state->near_edges;
  long long thr_lo = state->thr_lo, thr_hi = state->thr_hi;
  int pass = state->pass, status = 0;
  unsigned near_sel = state->near_sel, far_sel = state->far_sel, passes = 0, buckets = 0, ph = 0;
  unsigned long long relaxed_edges = 0;
  long long dlt = state->delta_cur;
  unsigned long long bwork = state->bucket_work;
  const unsigned long long light = state->light;
  unsigned light_run = state->light_run;
  const unsigned adapt_after = state->adapt_after;
  int32_t maxd = 0;
  bool over = false;
  auto clamp = [](long long x) { return (int32_t)(x > GDN_DIST_INF ? GDN_DIST_INF : x); };
  // every phase: its counter set, and the reset of the next phase's
  auto begin_phase = [&]() -> SsspCoopCnt * {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      SsspCoopCnt *nxt = cnt + ((ph + 1u) % 3u);
      __hip_atomic_store(&nxt->nn, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&nxt->nf, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&nxt->edges, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&nxt->min_far, (int)GDN_DIST_INF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return cnt + (ph % 3u);
  };
  for (;;) {
    if (n_near > 0) {
      // ---- one relax pass over NEAR
      SsspCoopCnt *cur = begin_phase();
      ++pass;
      ++passes;
      relaxed_edges += near_edges;
      bwork += near_edges;
      const vid_t *near_in = near_sel ? near1 : near0;
      vid_t *near_out = near_sel ? near0 : near1;
      vid_t *far_cur = far_sel ? far1 : far0;
      const int32_t lo = clamp(thr_lo), hi = clamp(thr_hi);
      unsigned long long edges = 0;
      auto relax = [&](bool valid, eoff_t k, int32_t du) {  // convergent: one edge per lane
        bool to_near = false, to_far = false;
        vid_t dst = 0;
        if (valid) {
          dst = colidx[k];
          const int32_t nd = du + weight[k];
          if (nd < __hip_atomic_load(dist + dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            const int32_t old = atomicMin(&dist[dst], nd);
            if (nd < old) {
              maxd = nd > maxd ? nd : maxd;
              if (nd < hi) to_near = atomicExch(&stamp[dst], pass) != pass;
              else to_far = atomicExch(&in_far[dst], 1u) == 0u;
            }
          }
        }
        const unsigned pn = sssp_global_slot(&cur->nn, to_near);
        if (to_near) {
          if (pn < cap) sssp_st(near_out + pn, dst);
          else over = true;
          edges += rowptr[dst + 1] - rowptr[dst];
        }
        const unsigned pf = sssp_global_slot(&cur->nf, to_far);
        if (to_far) {
          if (n_far + pf < cap) sssp_st(far_cur + n_far + pf, dst);
          else over = true;
        }
      };
      for (unsigned i0 = gt - lane; i0 < n_near; i0 += nt) {
        const unsigned i = i0 + lane;
        eoff_t b = 0, e = 0;
        int32_t du = 0;
        if (i < n_near) {
          const vid_t v = sssp_ld(near_in + i);
          du = __hip_atomic_load(dist + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (du >= lo) {  // else: settled in an earlier bucket (omp_base.cc:40)
            b = rowptr[v];
            e = rowptr[v + 1];
          }
        }
        const bool mine_long = e - b >= 64u;
        unsigned long long longs = __ballot(mine_long);
        while (longs) {
          const int leader = __ffsll((long long)longs) - 1;
          longs &= longs - 1ull;
          const eoff_t bb = __shfl(b, leader, 64), ee = __shfl(e, leader, 64);
          const int32_t dd = __shfl(du, leader, 64);
          for (eoff_t k0 = bb; k0 < ee; k0 += 64) relax(k0 + lane < ee, k0 + lane, dd);
        }
        if (mine_long) b = e;
        for (eoff_t j = 0; __any(b + j < e); j++) relax(b + j < e, b + j, du);
      }
      edges = gdn_wave_sum(edges);
      if (lane == 0 && edges) atomicAdd(&cur->edges, edges);
      if (__any(over) && lane == 0) __hip_atomic_store(&cur->over, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      gdn_grid_barrier(bar, gridDim.x);
      ph++;
      n_near = __hip_atomic_load(&cur->nn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      n_far += __hip_atomic_load(&cur->nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      near_edges = __hip_atomic_load(&cur->edges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      near_sel ^= 1u;
      if (__hip_atomic_load(&cur->over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        over = true;
        break;
      }
      if (n_near > 0) {
        if (n_near > max_v || near_edges > max_e) {
          status = 1;
          break;
        }
        continue;
      }
    }
    // ---- NEAR is empty: the next non-empty bucket (omp_base.cc:66-72)
    if (n_far == 0) {
      status = 0;
      break;
    }
    if (n_far > max_far) {
      status = 2;
      break;
    }
    // one pass per bucket change wherever the bucket behind the old one is not empty (see sssp_small_kernel)
    vid_t *near_in = near_sel ? near1 : near0;
    dlt = sssp_adapt_delta(dlt, (long long)delta, bwork, light, light_run, adapt_after);
    bwork = 0ull;
    long long spec_lo = thr_hi, spec_hi = thr_hi + dlt;
    for (;;) {
      vid_t *far_cur = far_sel ? far1 : far0, *far_nxt = far_sel ? far0 : far1;
      SsspCoopCnt *cur = begin_phase();
      const int32_t ohi = clamp(thr_hi), nhi = clamp(spec_hi);
      unsigned long long deg_sum = 0;
      int32_t dmin = GDN_DIST_INF;
      for (unsigned i0 = gt - lane; i0 < n_far; i0 += nt) {
        const unsigned i = i0 + lane;
        vid_t w = -1;
        int32_t dd = 0;
        if (i < n_far) {
          w = sssp_ld(far_cur + i);
          dd = __hip_atomic_load(dist + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool live = w >= 0, to_far = live && dd >= nhi, to_near = live && !to_far && dd >= ohi;
        if (live && !to_far) __hip_atomic_store(in_far + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned pf = sssp_global_slot(&cur->nf, to_far);
        if (to_far) {
          sssp_st(far_nxt + pf, w);
          dmin = dd < dmin ? dd : dmin;
        }
        const unsigned pn = sssp_global_slot(&cur->nn, to_near);
        if (to_near) {
          sssp_st(near_in + pn, w);
          deg_sum += rowptr[w + 1] - rowptr[w];
        }
      }
      deg_sum = gdn_wave_sum(deg_sum);
      if (lane == 0 && deg_sum) atomicAdd(&cur->edges, deg_sum);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int32_t t = __shfl_xor(dmin, o, 64);
        dmin = t < dmin ? t : dmin;
      }
      if (lane == 0 && dmin != GDN_DIST_INF) atomicMin(&cur->min_far, dmin);
      gdn_grid_barrier(bar, gridDim.x);
      ph++;
      const unsigned nn = __hip_atomic_load(&cur->nn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned nf = __hip_atomic_load(&cur->nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int32_t mn = __hip_atomic_load(&cur->min_far, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      near_edges = __hip_atomic_load(&cur->edges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      far_sel ^= 1u;
      n_far = nf;
      if (nn > 0 || nf == 0) {
        if (nn > 0) {
          thr_lo = spec_lo;
          thr_hi = spec_hi;
          ++buckets;
        }
        n_near = nn;
        break;
      }
      thr_hi = spec_hi;  // that bucket was empty and everything kept lies behind it
      spec_lo = ((long long)mn / dlt) * dlt;
      if (spec_lo < thr_hi) spec_lo = thr_hi;
      spec_hi = spec_lo + dlt;
    }
    if (n_near > max_v || near_edges > max_e) {
      status = 1;
      break;
    }
  }
  // largest distance written: one atomic per wave that raises it
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int32_t t = __shfl_xor(maxd, o, 64);
    maxd = t > maxd ? t : maxd;
  }
  if (lane == 0 && maxd > __hip_atomic_load(&state->max_dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&state->max_dist, maxd);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    state->n_near = n_near;
    state->n_far = n_far;
    state->near_edges = near_edges;
    state->thr_lo = thr_lo;
    state->thr_hi = thr_hi;
    state->pass = pass;
    state->status = status;
    state->near_sel = near_sel;
    state->far_sel = far_sel;
    state->overflow = over ? 1u : 0u;
    state->passes = passes;
    state->buckets = buckets;
    state->relaxed_edges = relaxed_edges;
    state->delta_cur = dlt;
    state->bucket_work = bwork;
    state->light_run = light_run;
  }
}

// improved-row bitmap -> vertex queue.  Persistent grid: a workgroup owns a contiguous range of words, counts its rows
// first and reserves its part of the queue with ONE atomic (one per wave on the hot counter cost 0.19 ms for 8192 waves),
// then writes the rows in order; also sums their out-degrees (the host's dense / worklist decision).
#define SSSP_B2Q_BLOCKS 512
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_bitmap_to_queue(const unsigned *__restrict__ bits, unsigned nwords, int32_t m, vid_t *__restrict__ q,
                     SsspCounters *cnt, unsigned cap, const eoff_t *__restrict__ rowptr) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK];
  __shared__ unsigned s_base;
  const unsigned per = (nwords + gridDim.x - 1) / gridDim.x;
  const unsigned w0 = blockIdx.x * per, w1 = w0 + per < nwords ? w0 + per : nwords;
  unsigned mine = 0;
  for (unsigned w = w0 + threadIdx.x; w < w1; w += GDN_BLOCK) mine += __popc(bits[w]);
  const unsigned tot = gdn_block_sum(mine, s_scan);
  if (tot == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(&cnt->near_count, tot);
  __syncthreads();
  unsigned base = s_base;
  unsigned long long deg = 0;
  for (unsigned t0 = w0; t0 < w1; t0 += GDN_BLOCK) {  // block-uniform trip count
    const unsigned w = t0 + threadIdx.x;
    unsigned word = (w < w1) ? bits[w] : 0u;
    unsigned total;
    unsigned pos = base + gdn_block_excl_scan((unsigned)__popc(word), s_scan, &total);
    base += total;
    while (word) {
      const int k = __ffs((int)word) - 1;
      word &= word - 1u;
      const unsigned v = w * 32u + (unsigned)k;
      if (v < (unsigned)m) {
        if (pos < cap) q[pos] = (vid_t)v;
        else cnt->overflow = 1u;
        if (rowptr) deg += rowptr[v + 1] - rowptr[v];
      }
      pos++;
    }
  }
  deg = gdn_wave_sum(deg);
  if (gdn_lane() == 0 && deg) atomicAdd(&cnt->relaxed, deg);
}

// ------------------------------------------------------------------------------------------
// BINNED relax pass: Bellman-Ford over a LIST of rows whose out-edges are a fraction of the graph -- what a dense sweep
// does for all edges (9.5 B per edge of the GRAPH, RMAT-24: 0.59 ms whatever improved), done for the list's edges only,
// and still without a random access per edge (a worklist pass pays a divergent probe + a global atomicMin per edge).
// Propagation blocking made on the fly, the binned top-down BFS level (gdn_bfs.hip, bfs_btd_*) with a payload:
//   sssp_bin_kernel / sssp_bin_big_kernel   expand the list's rows (gdn_expand.hpp); a wave step's (destination,
//       candidate) pairs are grouped by destination bin (ballot match), each group reserves room in its bin's list with
//       one atomic and writes its 8-byte entries side by side.  SSSP_BIN_SUB lists per bin, one per XCD.
//   sssp_bin_apply_kernel   one workgroup per bin: its lists are streamed once, candidates are min-ed into the bin's
//       2^logb LDS words, then the sweep's epilogue -- rows whose minimum beats their distance are written, marked in
//       the improved bitmap and counted with their out-edges.
// 8 B (colidx + weight) read + 8 B written + 8 B read per LIST edge.  A list that would overflow sets a flag, the apply
// kernel then only resets the counters, and the host runs a dense sweep instead (which needs nothing from the lists).
// ------------------------------------------------------------------------------------------
#define SSSP_BIN_SUB 8
#define SSSP_BIN_LOGB 15  // 2^15 rows per bin: 128 KB of LDS minima
#define SSSP_BIN_THREADS 1024
#define SSSP_BIN_FRAC 4   // a binned pass takes lists whose rows own at most nnz / 4 out-edges ...
#define SSSP_BIN_SLACK 3  // ... and every list has room for 3 times its even share
__device__ __forceinline__ unsigned sssp_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }  // HW_REG_XCC_ID
struct SsspBinVis {
  const vid_t *__restrict__ colidx;
  const int32_t *__restrict__ weight;
  const int32_t *__restrict__ dist;
  unsigned long long *__restrict__ buf;  // entries: candidate << 32 | destination
  unsigned *cur;                         // counter of list (bin, sub) at cur[(bin * SUB + sub) * 32]: a 128-byte line each
  unsigned *overflow;
  unsigned cap_each, sub;
  int bin_bits;
  int32_t du;  // per-lane: distance of this lane's row
  __device__ __forceinline__ void begin_big(vid_t v) { du = dist[v]; }
  __device__ __forceinline__ void edge(int owner, eoff_t k, bool valid) {
    const int32_t d_src = __shfl(du, owner, 64);
    vid_t dst = 0;
    unsigned bin = 0, cand = 0;
    if (valid) {
      dst = __builtin_nontemporal_load(colidx + k);
      cand = (unsigned)(d_src + __builtin_nontemporal_load(weight + k));
      bin = (unsigned)dst >> SSSP_BIN_LOGB;
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
    const size_t slot = (size_t)bin * SSSP_BIN_SUB + sub;
    unsigned base = 0;
    // the counter is only added to from ONE XCD (sub = its id) and read behind a kernel boundary: workgroup scope keeps
    // the add in that XCD's L2 (gdn_bfs.hip, bfs_btd_reserve)
    if (valid && rank == 0u) base = __hip_atomic_fetch_add(cur + slot * 32, (unsigned)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    base = __shfl(base, valid ? __ffsll((long long)peers) - 1 : (int)lane, 64);
    if (valid) {
      const unsigned pos = base + rank;
      if (pos < cap_each) buf[slot * cap_each + pos] = ((unsigned long long)cand << 32) | (unsigned)dst;
      else *overflow = 1u;
    }
  }
  __device__ __forceinline__ void finish() {}
};

__global__ void __launch_bounds__(GDN_BLOCK)
sssp_bin_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ inq, unsigned n, ExpBigList big, SsspBinVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vid_t v = 0;
  vis.du = 0;
  if (i < n) {
    v = inq[i];
    vis.du = vis.dist[v];
    b = rowptr[v];
    e = rowptr[v + 1];
  }
  vis.sub = sssp_xcc_id() & (SSSP_BIN_SUB - 1);
  gdn_expand_wave(b, e, v, big, vis, s_scan[threadIdx.x >> 6]);
}

__global__ void __launch_bounds__(GDN_BLOCK)
sssp_bin_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, SsspBinVis vis) {
  vis.du = 0;
  vis.sub = sssp_xcc_id() & (SSSP_BIN_SUB - 1);
  gdn_expand_big_items(rowptr, big, vis);
}

__global__ void __launch_bounds__(SSSP_BIN_THREADS)
sssp_bin_apply_kernel(const unsigned long long *__restrict__ buf, unsigned *cur, const unsigned *__restrict__ overflow,
                      unsigned cap_each, int32_t m, int32_t *__restrict__ dist, unsigned *__restrict__ improved_bits,
                      const eoff_t *__restrict__ rowptr, SsspCounters *cnt, vid_t *__restrict__ queue, unsigned qcap) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_min[];  // 2^SSSP_BIN_LOGB words
  __shared__ unsigned long long s_red[SSSP_BIN_THREADS / 64];
  __shared__ int s_max[SSSP_BIN_THREADS / 64];
  const unsigned bin = blockIdx.x, bn = 1u << SSSP_BIN_LOGB;
  const bool skip = *overflow != 0u;  // the lists are incomplete: the host runs a dense sweep instead
  if (!skip)
    for (unsigned i = threadIdx.x; i < bn; i += SSSP_BIN_THREADS) s_min[i] = (unsigned)GDN_DIST_INF;
  __syncthreads();
  for (unsigned sub = 0; sub < SSSP_BIN_SUB && !skip; sub++) {
    const size_t slot = (size_t)bin * SSSP_BIN_SUB + sub;
    unsigned n = cur[slot * 32];
    n = n < cap_each ? n : cap_each;
    const unsigned long long *__restrict__ src = buf + slot * cap_each;
    constexpr int UNR = 4;
    for (unsigned i0 = threadIdx.x; i0 < n; i0 += UNR * SSSP_BIN_THREADS) {
      unsigned long long e[UNR];
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const unsigned i = i0 + (unsigned)r * SSSP_BIN_THREADS;
        e[r] = i < n ? __builtin_nontemporal_load(src + i) : ~0ull;
      }
#pragma unroll
      for (int r = 0; r < UNR; r++)
        if (e[r] != ~0ull) atomicMin(&s_min[(unsigned)e[r] & (bn - 1u)], (unsigned)(e[r] >> 32));
    }
  }
  __syncthreads();
  if (threadIdx.x < SSSP_BIN_SUB) cur[((size_t)bin * SSSP_BIN_SUB + threadIdx.x) * 32] = 0u;  // ready for the next pass
  if (skip) return;
  // epilogue: one row per thread and step (a wave covers 64 consecutive rows = 2 bitmap words); the improved rows also go
  // into the queue of the next pass: a thread remembers its 32 steps' outcomes in one mask, the workgroup reserves its
  // share of the queue with ONE atomic, the second walk places the rows
  static_assert((1u << SSSP_BIN_LOGB) / SSSP_BIN_THREADS == 32u, "one mask bit per step");
  const unsigned lane = gdn_lane(), w = threadIdx.x >> 6;
  const size_t row0 = (size_t)bin << SSSP_BIN_LOGB;
  unsigned mine = 0u, wave_cnt = 0u;
  unsigned long long edges = 0;
  int mx = 0;
#pragma unroll 4
  for (unsigned k = 0; k < 32u; k++) {
    const unsigned i = k * SSSP_BIN_THREADS + threadIdx.x;
    const size_t row = row0 + i;
    bool imp = false;
    const unsigned nm = s_min[i];
    if (row < (size_t)m && nm != (unsigned)GDN_DIST_INF) {
      const unsigned old = (unsigned)dist[row];
      if (nm < old) {
        dist[row] = (int32_t)nm;
        mx = (int)nm > mx ? (int)nm : mx;
        edges += rowptr[row + 1] - rowptr[row];
        imp = true;
      }
    }
    const unsigned long long mask = __ballot(imp);
    if ((lane & 31u) == 0 && row < (((size_t)m + 31) & ~(size_t)31)) improved_bits[row >> 5] = (unsigned)(mask >> (lane & 32u));
    mine |= imp ? (1u << k) : 0u;
    wave_cnt += (unsigned)__popcll(mask);
  }
  edges = gdn_wave_sum(edges);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(mx, o, 64);
    mx = t > mx ? t : mx;
  }
  __shared__ unsigned s_wcnt[SSSP_BIN_THREADS / 64], s_base;
  if (lane == 0) {
    s_wcnt[w] = wave_cnt;
    s_red[w] = edges;
    s_max[w] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long ed = 0;
    unsigned a = 0;
    int mm = 0;
    for (int i = 0; i < SSSP_BIN_THREADS / 64; i++) {
      a += s_wcnt[i];
      ed += s_red[i];
      mm = s_max[i] > mm ? s_max[i] : mm;
    }
    s_base = 0u;
    if (a) {
      s_base = atomicAdd(&cnt->near_count, a);  // the queue of the next pass
      atomicAdd(&cnt->relaxed, (unsigned long long)a);  // rows improved (as the sweep's epilogue counts them)
      atomicAdd(&cnt->improved_edges, ed);              // their out-edges
      atomicMax(&cnt->max_dist, mm);
    }
  }
  __syncthreads();
  unsigned pos = s_base;
  for (unsigned i = 0; i < w; i++) pos += s_wcnt[i];
  for (unsigned k = 0; k < 32u && wave_cnt; k++) {  // wave_cnt is wave-uniform
    const bool imp = (mine >> k) & 1u;
    const unsigned long long mask = __ballot(imp);
    if (imp) {
      const unsigned at = pos + (unsigned)__popcll(mask & gdn_lanemask_lt());
      if (at < qcap) queue[at] = (vid_t)(row0 + k * SSSP_BIN_THREADS + threadIdx.x);
      else cnt->overflow = 1u;
    }
    pos += (unsigned)__popcll(mask);
  }
}

// out-degree sum of the rows a bitmap marks (what a frontier-proportional pass over them would relax)
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_bitmap_edges_kernel(const unsigned *__restrict__ bits, unsigned nwords, int32_t m, const eoff_t *__restrict__ rowptr,
                         unsigned long long *__restrict__ out) {
  unsigned long long deg = 0;
  for (unsigned w = blockIdx.x * GDN_BLOCK + threadIdx.x; w < nwords; w += gridDim.x * GDN_BLOCK) {
    unsigned word = bits[w];
    while (word) {
      const unsigned v = w * 32u + (unsigned)(__ffs((int)word) - 1);
      word &= word - 1u;
      if (v < (unsigned)m) deg += rowptr[v + 1] - rowptr[v];
    }
  }
  deg = gdn_wave_sum(deg);
  if (gdn_lane() == 0 && deg) atomicAdd(out, deg);
}

struct gdn_sssp_plan {
  const gdn_graph *g = nullptr;
  const int32_t *d_weight = nullptr;
  bool dense = false;
  PbPlan pb;               // of the OUT-CSR (rows are sources), no fp32 vals
  DevBuf<float> Wp;        // weights in tile order (int32 bits); released when a narrower copy serves
  DevBuf<uint8_t> Wn;      // the same as u8 / u16 (w_bytes 1 / 2)
  int w_bytes = 4;         // 0: all weights equal (w_min), no stream
  int32_t w_min = 0, w_max = 0;
  // binned relax passes (sssp_bin_*): nbins x SSSP_BIN_SUB lists of bin_cap_each 8-byte entries, their counters, the flag
  DevBuf<unsigned long long> bin_buf;
  DevBuf<unsigned> bin_cur, bin_ovf;
  unsigned bin_nbins = 0, bin_cap_each = 0;
  int bin_bits = 0;
  // RECORD TIERS of the sweeps (sssp_build_tiers): the out-edges of the sources of highest out-degree leave the blocked
  // layout; phase B reads them as 4-byte (source index << 14 | row) records + a 1-byte weight, the source's distance comes
  // from a per-sweep table (tier_tab, L2 resident) -- 5 B per edge instead of the 9.5 B of an edge that travels through cand
  int n_tiers = 0;
  unsigned tier_off[SSSP_MAX_TIERS + 1] = {};  // first source of tier t in tier_ids / tier_tab
  DevBuf<uint32_t> tier_ids;                   // the tier sources (descending out-degree)
  DevBuf<unsigned> tier_tab;                   // their distances, refreshed per sweep
  DevBuf<uint32_t> tier_rec;                   // records, tier-major then bin-major
  DevBuf<uint8_t> tier_w;                      // their weights (w_bytes 1; none when all weights are equal)
  DevBuf<eoff_t> tier_ptr;                     // n_tiers x nbins + 1 offsets into tier_rec
  DevBuf<uint32_t> tier_cnt;                   // interleaved streams only (SsspTierArgs::cnt): records per stream
  unsigned long long tier_edges = 0;
  DevBuf<unsigned> cand;   // candidate distances, bin-major (u8, u16 or u32 per sweep)
  int cand_bits = 0;       // width of the candidates the last sweep wrote (0: none yet, every byte is 0xFF).  The slots in the
                           // alignment gaps are never written and must read as "no path" (all ones): a sweep of another
                           // width finds the bytes of the old width's REAL candidates there -- the buffer is wiped first
  DevBuf<unsigned> improved;
  DevBuf<unsigned> bad;    // 1 word: a 16-bit candidate overflowed (cannot happen; checked)
  DevBuf<SsspSmallState> small;
  DevBuf<SsspCoopCnt> coop_cnt;  // 3 rotating counter sets of sssp_coop_kernel
  DevBuf<unsigned> coop_bar;     // its grid barrier
  int coop_blocks = 0;           // 0: no cooperative launches
  GdnMailbox mail;               // the per-phase read back of the counters
  DevBuf<vid_t> near0, near1, far0, far1;
  DevBuf<int32_t> stamp;
  DevBuf<unsigned> in_far;
  DevBuf<unsigned long long> bigitems;
  DevBuf<SsspCounters> cnt;
  unsigned cap = 0, bigcap = 0, nwords = 0;
  double prep_ms = 0;
  // ALL WEIGHTS EQUAL (the reference main's own input: src/sssp/main.cc:26 fills 1): shortest distances are hop counts times
  // that weight, so a resident plan solves through the direction-optimising BFS plan (gdn_bfs.hip) on the transpose it
  // builds once -- no bucket, no relaxation is repeated -- and converts depths to distances (sssp_depth_to_dist_kernel)
  gdn_graph *bfs_gin = nullptr;
  gdn_bfs_plan *bfs = nullptr;
  int32_t bfs_w = 0;
  gdn_sssp_plan() {}
  gdn_sssp_plan(const gdn_sssp_plan &) = delete;
  gdn_sssp_plan &operator=(const gdn_sssp_plan &) = delete;
  ~gdn_sssp_plan() {
    if (bfs) gdn_bfs_plan_free(bfs);
    if (bfs_gin) gdn_graph_free(bfs_gin);
  }
};

// depth (MYINFINITY = unreached) -> distance = depth x w (kDistInf = unreached, and beyond the int range like the relax kernels)
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_depth_to_dist_kernel(int32_t *__restrict__ dist, int32_t m, int32_t w) {
  const int32_t v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v >= m) return;
  const int32_t d = dist[v];
  const long long nd = (long long)d * (long long)w;
  dist[v] = (d == GDN_MYINFINITY || nd >= (long long)GDN_DIST_INF) ? GDN_DIST_INF : (int32_t)nd;
}

// ------------------------------------------------------------------------------------------
// RECORD TIERS of the dense sweeps.  PageRank's record tiers (gdn_pb.hpp) for a min-plus sweep: an edge of the blocked layout
// costs 9.5 B per sweep (U 2 + G 0.5 + weight 1 + candidate 2 written, candidate 2 + V 2 read); an edge that leaves a source
// of high out-degree is instead kept as a RECORD in the order phase B wants it (bin-major): 4 bytes (source index << 14 | row
// in the bin; 15 row bits) + 1 byte of weight, and the source's distance is looked up in a table refreshed per sweep (32 K entries for the
// first tier, up to 256 K for the others: L2 resident).  Sources are ranked by out-degree; those with at least 1/16 edge per
// bin (and 8) go into up to SSSP_MAX_TIERS tiers.  Needs weights of at most 8 bits (or all equal) and bins of at most 2^15 rows.
// ------------------------------------------------------------------------------------------
int gdn_radix_sort_u64(unsigned long long *a, unsigned long long *b, unsigned long long n, unsigned begin_bit, unsigned end_bit,
                       const unsigned long long **sorted);

static inline unsigned sssp_tier_first_host(unsigned t) { return t == 0 ? 0u : SSSP_TIER0 + (t - 1u) * SSSP_TIER_N; }
__device__ __forceinline__ unsigned sssp_tier_of(unsigned r) { return r < SSSP_TIER0 ? 0u : 1u + (r - SSSP_TIER0) / SSSP_TIER_N; }
__device__ __forceinline__ unsigned sssp_tier_first(unsigned t) { return t == 0 ? 0u : SSSP_TIER0 + (t - 1u) * SSSP_TIER_N; }

__global__ void __launch_bounds__(GDN_BLOCK)
sssp_tier_degkeys_kernel(const eoff_t *__restrict__ rowptr, int32_t m, unsigned long long *__restrict__ keys) {
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (unsigned)m) {
    const eoff_t d = rowptr[v + 1] - rowptr[v];
    keys[v] = ((unsigned long long)(d > 0xFFFFFFFFull ? 0xFFFFFFFFull : d) << 32) | v;
  }
}
// number of keys (ascending by degree) whose degree is >= min_deg
__global__ void sssp_tier_count_kernel(const unsigned long long *__restrict__ sorted, int32_t m, unsigned min_deg, unsigned *out) {
  if (threadIdx.x || blockIdx.x) return;
  size_t lo = 0, hi = (size_t)m;  // first index with degree >= min_deg
  while (lo < hi) {
    const size_t mid = (lo + hi) >> 1;
    if ((unsigned)(sorted[mid] >> 32) < min_deg) lo = mid + 1;
    else hi = mid;
  }
  *out = (unsigned)((size_t)m - lo);
}
// rank r (0 = the largest out-degree) -> ids[r], degs[r], cls[id] = 1 + its tier
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_tier_assign_kernel(const unsigned long long *__restrict__ sorted, int32_t m, unsigned n_ts, uint8_t *__restrict__ cls,
                        uint32_t *__restrict__ ids, uint32_t *__restrict__ degs) {
  const unsigned r = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (r >= n_ts) return;
  const unsigned long long key = sorted[(size_t)m - 1 - r];
  const unsigned id = (unsigned)(key & 0xFFFFFFFFull);
  ids[r] = id;
  degs[r] = (unsigned)(key >> 32);
  cls[id] = (uint8_t)(1u + sssp_tier_of(r));
}
// one wave per tier source: its out-edges as keys  (tier * nbins + bin) << 40 | record << 8 | weight
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_tier_keys_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const int32_t *__restrict__ weight,
                      const uint32_t *__restrict__ ids, const eoff_t *__restrict__ offs, unsigned n_ts, unsigned nbins, int lb,
                      unsigned long long *__restrict__ keys) {
  const unsigned lane = gdn_lane();
  const size_t nwaves = ((size_t)gridDim.x * GDN_BLOCK) >> 6;
  for (size_t r = ((size_t)blockIdx.x * GDN_BLOCK + threadIdx.x) >> 6; r < (size_t)n_ts; r += nwaves) {
    const unsigned id = ids[r], t = sssp_tier_of((unsigned)r), idx = (unsigned)r - sssp_tier_first(t);
    const eoff_t b = rowptr[id], e = rowptr[id + 1], o = offs[r];
    for (eoff_t k = b + lane; k < e; k += 64) {
      const unsigned dst = (unsigned)colidx[k];
      const unsigned long long tb = (unsigned long long)t * nbins + (dst >> lb);
      const unsigned rec = (idx << SSSP_TIER_ROW_BITS) | (dst & ((1u << lb) - 1u));
      keys[o + (k - b)] = (tb << 40) | ((unsigned long long)rec << 8) | ((unsigned)weight[k] & 0xFFu);
    }
  }
}
// sorted keys -> records, weights, and ptr[x] = first position whose (tier, bin) index is >= x (ptr[n_tb] = n)
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_tier_split_kernel(const unsigned long long *__restrict__ sorted, unsigned long long n, unsigned n_tb,
                       uint32_t *__restrict__ rec, uint8_t *__restrict__ w8, eoff_t *__restrict__ ptr) {
  const unsigned long long j = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (j >= n) return;
  const unsigned long long key = sorted[j];
  rec[j] = (uint32_t)(key >> 8);
  if (w8) w8[j] = (uint8_t)(key & 0xFFull);
  const long long tb = (long long)(key >> 40), prev = j ? (long long)(sorted[j - 1] >> 40) : -1ll;
  for (long long x = prev + 1; x <= tb; x++) ptr[x] = j;
  if (j == n - 1)
    for (long long x = tb + 1; x <= (long long)n_tb; x++) ptr[x] = n;
}
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_tier_gather_kernel(const int32_t *__restrict__ dist, const uint32_t *__restrict__ ids, unsigned n, unsigned *__restrict__ tab) {
  const unsigned k = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (k < n) tab[k] = (unsigned)dist[ids[k]];
}

// picks the tier sources and builds their record streams; cls (one byte per source: 0 = stays in the blocked layout) is what
// pb_build filters the blocked layout with.  No tiers (n_tiers = 0, cls empty) when nothing qualifies.
static int sssp_build_tiers(gdn_sssp_plan &p, const gdn_graph *g, const int32_t *d_weight, int lb, DevBuf<uint8_t> &cls) {
  const int32_t m = g->m;
  const unsigned nbins = (unsigned)(((uint64_t)m + (1u << lb) - 1) >> lb);  // the bins of the blocked layout (lb <= 14)
  // floor: a quarter of an edge per bin (RMAT-24, 1024 bins: 256 out-edges -> two tiers, 64.5 % of the edges, 3.47 -> 3.15 ms;
  // 64: three tiers, 80 %, 3.33 -- phase B is bound by the issue side of the memory pipeline then, not by bytes;
  // profiles/r03_sssp_tiers.txt)
  unsigned min_deg = nbins / 4u < 8u ? 8u : nbins / 4u;
  if (const char *e = gdn_test_option("GDN_SSSP_TIER_MIN_DEG")) min_deg = atoi(e) > 0 ? (unsigned)atoi(e) : min_deg;  // test / tuning knob
  int max_tiers = SSSP_MAX_TIERS;
  if (const char *e = gdn_test_option("GDN_SSSP_TIERS")) max_tiers = atoi(e) < SSSP_MAX_TIERS ? atoi(e) : SSSP_MAX_TIERS;
  if (max_tiers <= 0 || m < 2) return GDN_OK;
  DevBuf<unsigned long long> ka, kb;
  GDN_TRY(ka.alloc((size_t)m));
  GDN_TRY(kb.alloc((size_t)m));
  hipLaunchKernelGGL(sssp_tier_degkeys_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, ka.p);
  const unsigned long long *sorted = nullptr;
  GDN_TRY(gdn_radix_sort_u64(ka.p, kb.p, (unsigned long long)m, 32u, 64u, &sorted));
  DevBuf<unsigned> cntb;
  GDN_TRY(cntb.alloc(1));
  hipLaunchKernelGGL(sssp_tier_count_kernel, dim3(1), dim3(64), 0, 0, sorted, m, min_deg, cntb.p);
  unsigned n_ts = 0;
  GDN_HIP(hipMemcpy(&n_ts, cntb.p, 4, hipMemcpyDeviceToHost));
  const unsigned cap = SSSP_TIER0 + (unsigned)(max_tiers - 1) * SSSP_TIER_N;
  if (n_ts > cap) n_ts = cap;
  if (n_ts == 0) return GDN_OK;
  int nt = 1;
  while (nt < max_tiers && sssp_tier_first_host((unsigned)nt) < n_ts) nt++;
  GDN_TRY(cls.alloc((size_t)m));
  GDN_HIP(hipMemset(cls.p, 0, (size_t)m));
  DevBuf<uint32_t> degs;
  DevBuf<eoff_t> offs;
  GDN_TRY(p.tier_ids.alloc(n_ts));
  GDN_TRY(degs.alloc((size_t)n_ts + 1));
  GDN_TRY(offs.alloc((size_t)n_ts + 1));
  GDN_HIP(hipMemset(degs.p, 0, ((size_t)n_ts + 1) * 4));
  hipLaunchKernelGGL(sssp_tier_assign_kernel, dim3(gdn_nblocks(n_ts)), dim3(GDN_BLOCK), 0, 0, sorted, m, n_ts, cls.p, p.tier_ids.p,
                     degs.p);
  GDN_TRY(gdn_exclusive_scan_u32_to_u64(degs.p, offs.p, (size_t)n_ts, 0));  // (writes n_ts + 1 offsets: offs[n_ts] = all edges)
  eoff_t n_e = 0;
  GDN_HIP(hipMemcpy(&n_e, offs.p + n_ts, sizeof(eoff_t), hipMemcpyDeviceToHost));
  if (n_e == 0) {
    cls.release();
    p.tier_ids.release();
    return GDN_OK;
  }
  ka.release();  // (sorted may live in either buffer: the assign kernel is done with it)
  GDN_HIP(hipDeviceSynchronize());
  kb.release();
  DevBuf<unsigned long long> ea, eb;
  GDN_TRY(ea.alloc((size_t)n_e));
  GDN_TRY(eb.alloc((size_t)n_e));
  hipLaunchKernelGGL(sssp_tier_keys_kernel, dim3(4096), dim3(GDN_BLOCK), 0, 0, g->rowptr, g->colidx, d_weight, p.tier_ids.p, offs.p, n_ts,
                     nbins, lb, ea.p);
  const unsigned n_tb = (unsigned)nt * nbins;
  unsigned tb_bits = 1;
  while ((1ull << tb_bits) < n_tb) tb_bits++;
  const unsigned long long *es = nullptr;
  GDN_TRY(gdn_radix_sort_u64(ea.p, eb.p, n_e, 40u, 40u + tb_bits, &es));
  GDN_TRY(p.tier_rec.alloc((size_t)n_e + 16));
  if (p.w_bytes == 1) GDN_TRY(p.tier_w.alloc((size_t)n_e + 16));
  GDN_TRY(p.tier_ptr.alloc((size_t)n_tb + 1));
  hipLaunchKernelGGL(sssp_tier_split_kernel, dim3(gdn_nblocks(n_e)), dim3(GDN_BLOCK), 0, 0, es, n_e, n_tb, p.tier_rec.p,
                     p.w_bytes == 1 ? p.tier_w.p : nullptr, p.tier_ptr.p);
  GDN_TRY(p.tier_tab.alloc(n_ts));
  GDN_HIP(hipGetLastError());
  GDN_HIP(hipDeviceSynchronize());
  p.n_tiers = nt;
  for (int t = 0; t <= nt; t++) {
    const unsigned f = sssp_tier_first_host((unsigned)t);
    p.tier_off[t] = f < n_ts ? f : n_ts;
  }
  p.tier_edges = n_e;
  if (gdn_option("GDN_SSSP_TRACE"))
    fprintf(stderr, "[sssp] plan: %d record tiers, %u sources of >= %u out-edges, %llu edges (%.1f %% of the graph)\n", nt, n_ts, min_deg,
            (unsigned long long)n_e, 100.0 * (double)n_e / (double)(g->nnz ? g->nnz : 1));
  return GDN_OK;
}

static int sssp_plan_init(gdn_sssp_plan &p, const gdn_graph *g, const int32_t *d_weight, bool dense, bool bins = true) {
  HostTimer t;
  t.start();
  p.g = g;
  p.d_weight = d_weight;
  const int32_t m = g->m;
  p.cap = (unsigned)m;
  const uint64_t bigcap64 = g->nnz / EXP_CHUNK + (uint64_t)m / 64 + 1024;
  p.bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
  GDN_TRY(p.near0.alloc(p.cap));
  GDN_TRY(p.near1.alloc(p.cap));
  GDN_TRY(p.far0.alloc(p.cap));
  GDN_TRY(p.far1.alloc(p.cap));
  GDN_TRY(p.stamp.alloc(m));
  GDN_TRY(p.in_far.alloc(m));
  GDN_TRY(p.bigitems.alloc(p.bigcap));
  GDN_TRY(p.cnt.alloc(1));
  GDN_TRY(p.small.alloc(1));
  {  // the cooperative kernel: one workgroup per CU if the device takes cooperative launches
    int dev = 0, coop = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) == hipSuccess &&
        coop && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sssp_coop_kernel, SSSP_COOP_THREADS, 0) == hipSuccess && per_cu >= 1) {
      p.coop_blocks = cus;
      GDN_TRY(p.coop_cnt.alloc(3));
      GDN_TRY(p.coop_bar.alloc(GDN_GBAR_WORDS));
    }
    (void)hipGetLastError();
  }
  p.mail.init();
  if (dense && g->nnz > 0) {
    int lg = 10;
    while (lg < 15 && ((int64_t)1 << (lg + 9)) < (int64_t)m) lg++;
    if (const char *e = gdn_xoption("GDN_SSSP_LOG")) lg = atoi(e) >= 10 && atoi(e) <= 15 ? atoi(e) : lg;  // tuning knob
    // tiles padded so that a tile's candidates are whole 128-byte lines (a line shared by two tiles is written by two
    // workgroups at different times): 128 edges (u8 candidates) where tiles are long, 32 where the padding
    // would cost more than the partial lines
    // the weight range first: the record tiers keep a weight in 8 bits
    {
      DevBuf<int32_t> rng;
      GDN_TRY(rng.alloc(2));
      const int32_t init[2] = {0x7FFFFFFF, -0x7FFFFFFF - 1};
      GDN_HIP(hipMemcpy(rng.p, init, 8, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(sssp_weight_range_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, d_weight, (size_t)g->nnz, rng.p);
      int32_t h[2];
      GDN_HIP(hipMemcpy(h, rng.p, 8, hipMemcpyDeviceToHost));
      p.w_min = h[0];
      p.w_max = h[1];
      const char *e = gdn_test_option("GDN_SSSP_WBYTES");  // test / measurement knob: 4 keeps the int32 stream
      const int force = e ? atoi(e) : -1;
      if (p.w_min < 0) p.w_bytes = 4;  // negative weights: not narrowed (the solvers assume none, like the reference)
      else if (p.w_min == p.w_max && force < 0) p.w_bytes = 0;
      else if (p.w_max < 256 && (force < 0 || force == 1)) p.w_bytes = 1;
      else if (p.w_max < 65536 && (force < 0 || force == 2)) p.w_bytes = 2;
      else p.w_bytes = 4;
    }
    // record tiers: from 2^22 edges on, weights of at most 8 bits
    DevBuf<uint8_t> cls;
    int lb = lg;
    unsigned long long tiers_from = 1ull << 22;
    if (const char *e = gdn_test_option("GDN_SSSP_TIER_MIN_NNZ")) tiers_from = strtoull(e, nullptr, 10);  // (tests)
    const bool want_tiers = p.w_bytes <= 1 && p.w_min >= 0 && g->nnz >= tiers_from;
    bool built = false;
    {
      // round 4: the blocked layout and the record tiers from one counting pass and one two-level split over the edges
      // (pb_build_out_tiered, gdn_pbtier.hpp); GDN_PB_BUILDER=old: pb_build's sort of 8-byte keys + sssp_build_tiers
      const char *be = gdn_option("GDN_PB_BUILDER"), *pe = gdn_test_option("GDN_SSSP_PAD");
      if (!(be && be[0] == 'o')) {
        PbOutArgs oa;
        PbOutTiers ot;
        oa.g = g;
        oa.weight = d_weight;
        oa.log_chunk = lg;
        oa.log_bin = lb;
        oa.pad = pe ? (unsigned)atoi(pe) : 0u;  // 0: by the average tile, see below
        oa.log_group = 3;
        const unsigned nbins = (unsigned)(((uint64_t)m + (1u << lb) - 1) >> lb);
        unsigned min_deg = nbins / 4u < 8u ? 8u : nbins / 4u;  // a quarter of an edge per bin (profiles/r03_sssp_tiers.txt)
        if (const char *e = gdn_test_option("GDN_SSSP_TIER_MIN_DEG")) min_deg = atoi(e) > 0 ? (unsigned)atoi(e) : min_deg;
        int max_tiers = SSSP_MAX_TIERS;
        if (const char *e = gdn_test_option("GDN_SSSP_TIERS")) max_tiers = atoi(e) < SSSP_MAX_TIERS ? atoi(e) : SSSP_MAX_TIERS;
        oa.max_tiers = (want_tiers && m >= 2 && max_tiers > 0) ? max_tiers : 0;
        oa.tier_min_deg = min_deg;
        oa.caps[0] = SSSP_TIER0;
        for (int t = 1; t < PB_MAX_REC_TIERS; t++) oa.caps[t] = SSSP_TIER_N;
        oa.want_w8 = p.w_bytes == 1;
        {  // lane-interleaved record streams (a quarter of phase B's record and weight loads); GDN_SSSP_REC_IL=0: plain
          const char *ie = gdn_test_option("GDN_SSSP_REC_IL");
          oa.interleave = !(ie && ie[0] == '0');
        }
        const int rc = pb_build_out_tiered_run(oa, p.pb, p.Wp, ot);
        if (rc < 0) return rc;
        if (rc == GDN_OK) {
          built = true;
          p.n_tiers = ot.n;
          if (ot.n) {
            for (int t = 0; t <= ot.n; t++) p.tier_off[t] = ot.off[t];
            p.tier_ids.take(ot.ids);
            p.tier_rec.take(ot.rec);
            if (p.w_bytes == 1) p.tier_w.take(ot.w8);
            p.tier_ptr.take(ot.ptr);
            if (ot.interleaved) p.tier_cnt.take(ot.cnt);
            GDN_TRY(p.tier_tab.alloc(ot.off[ot.n]));
            p.tier_edges = ot.edges;
            if (gdn_option("GDN_SSSP_TRACE"))
              fprintf(stderr, "[sssp] plan: %d record tiers, %u sources of >= %u out-edges, %llu edges (%.1f %% of the graph)\n", ot.n,
                      ot.off[ot.n], min_deg, (unsigned long long)ot.edges, 100.0 * (double)ot.edges / (double)(g->nnz ? g->nnz : 1));
          }
        }
      }
    }
    if (!built) {
      if (want_tiers) {
        const int lbt = lg < SSSP_TIER_ROW_BITS ? lg : SSSP_TIER_ROW_BITS;  // a record keeps its row in 15 bits
        GDN_TRY(sssp_build_tiers(p, g, d_weight, lbt, cls));
        if (p.n_tiers) lb = lbt;
      }
      const double avg_tile = (double)(g->nnz - p.tier_edges) / ((double)(((uint64_t)m >> lg) + 1) * (double)(((uint64_t)m >> lb) + 1));
      // (64 against 32: RMAT-25 8.4 against 8.7 ms, RMAT-26 12.7 / 13.3, RMAT-27 -- 127 edges per tile -- 25.2 / 26.6,
      // profiles/r03_sssp_layout_knobs.txt)
      unsigned pad = avg_tile >= 1024.0 ? 128u : avg_tile >= 96.0 ? 64u : 32u;
      if (const char *e = gdn_test_option("GDN_SSSP_PAD")) pad = (unsigned)atoi(e);
      GDN_TRY(pb_build(g, m, lg, lb, p.pb, /*alloc_vals=*/false, reinterpret_cast<const float *>(d_weight), &p.Wp,
                       /*compact=*/false, /*rows_are_sources=*/true, pad, /*log_group=*/3, p.n_tiers ? cls.p : nullptr, 0));
    }
    GDN_TRY(p.cand.alloc(p.pb.n_pad + 8));
    GDN_TRY(p.bad.alloc(1));
    GDN_HIP(hipMemset(p.bad.p, 0, 4));
    // slots in the alignment gaps of the layout are never written by phase A: keep them neutral (all ones = INF of both
    // candidate widths)
    GDN_TRY(gdn_fill_i32(reinterpret_cast<int32_t *>(p.cand.p), -1, (size_t)p.pb.n_pad + 8, 0));
    // the weight stream as narrow as the weights allow
    {
      const size_t n = (size_t)p.pb.n_pad;
      if (p.w_bytes == 1 || p.w_bytes == 2) {
        GDN_TRY(p.Wn.alloc(n * (size_t)p.w_bytes + 64));
        if (p.w_bytes == 1)
          hipLaunchKernelGGL(HIP_KERNEL_NAME(sssp_narrow_weights_kernel<uint8_t>), dim3(4096), dim3(GDN_BLOCK), 0, 0,
                             reinterpret_cast<const uint32_t *>(p.Wp.p), n, p.Wn.p);
        else
          hipLaunchKernelGGL(HIP_KERNEL_NAME(sssp_narrow_weights_kernel<uint16_t>), dim3(4096), dim3(GDN_BLOCK), 0, 0,
                             reinterpret_cast<const uint32_t *>(p.Wp.p), n, reinterpret_cast<uint16_t *>(p.Wn.p));
        GDN_HIP(hipDeviceSynchronize());
      }
      if (p.w_bytes != 4) p.Wp.release();
    }
    p.nwords = (unsigned)(((uint64_t)p.pb.nbins << lb) / 32u);
    GDN_TRY(p.improved.alloc(p.nwords + 64));
    // OFF by default (GDN_SSSP_BINS=1 builds the lists and takes the passes): measured on RMAT-24, U[1,255], delta 16
    // (profiles/r03_sssp_binned_passes.txt) a binned pass costs ~0.19 ms + 39 ps per list edge -- 2.36 ms for the 55 M
    // out-edges of the 3.6 M rows the third sweep improved, where the fourth SWEEP takes 0.61 ms for all 263 M edges, and
    // no better than the worklist pass with its global atomicMin per edge (31 ps).  With 512 bins a wave step's 64
    // destinations fall into ~60 different bins: one list reservation PER EDGE, each a dependent round trip through the
    // L2 in front of the store -- the binned BFS level got its 6 ps per edge from hub rows walked 256 edges at a time
    // with four reservations in flight, and a low-degree list has none of that.  3.65 -> 5.9 ms with the passes on.
    const char *be = gdn_test_option("GDN_SSSP_BINS");
    if (bins && be && be[0] == '1') {
      // lists of the binned relax passes: a pass runs while the improved rows own at most nnz / SSSP_BIN_FRAC out-edges;
      // room for SSSP_BIN_SLACK times the even share per list (destinations that crowd into few bins overflow a list:
      // the pass is then repeated as a dense sweep)
      const uint64_t nb = (((uint64_t)m + (1u << SSSP_BIN_LOGB) - 1) >> SSSP_BIN_LOGB);
      uint64_t per = (g->nnz / SSSP_BIN_FRAC) * SSSP_BIN_SLACK / (nb * SSSP_BIN_SUB) + 1;
      per = per < 4096 ? 4096 : per;
      if (const char *e = gdn_test_option("GDN_SSSP_BIN_CAP")) per = atoi(e) > 0 ? (uint64_t)atoi(e) : per;  // test knob: short lists overflow
      per = (per + 15) & ~(uint64_t)15;
      if (per < 0x7FFFFFFFull && m >= (1 << SSSP_BIN_LOGB)) {
        GDN_TRY(p.bin_buf.alloc((size_t)(nb * SSSP_BIN_SUB * per)));
        GDN_TRY(p.bin_cur.alloc((size_t)(nb * SSSP_BIN_SUB * 32)));
        GDN_HIP(hipMemset(p.bin_cur.p, 0, (size_t)(nb * SSSP_BIN_SUB * 32) * 4));
        p.bin_nbins = (unsigned)nb;
        p.bin_cap_each = (unsigned)per;
        p.bin_bits = 0;
        while ((1ull << p.bin_bits) < nb) p.bin_bits++;
        if (hipFuncSetAttribute((const void *)sssp_bin_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                4 << SSSP_BIN_LOGB) != hipSuccess) {
          gdn_set_error("hipFuncSetAttribute(dynamic LDS, sssp_bin_apply_kernel)");
          return GDN_ERR_HIP;
        }
      }
    }
    const int lds = (int)((sizeof(unsigned) << lg) + 16);
    const void *fns[] = {(const void *)sssp_pb_expand_kernel<0, uint16_t>, (const void *)sssp_pb_expand_kernel<1, uint16_t>,
                         (const void *)sssp_pb_expand_kernel<2, uint16_t>, (const void *)sssp_pb_expand_kernel<4, uint16_t>,
                         (const void *)sssp_pb_expand_kernel<0, uint32_t>, (const void *)sssp_pb_expand_kernel<1, uint32_t>,
                         (const void *)sssp_pb_expand_kernel<2, uint32_t>, (const void *)sssp_pb_expand_kernel<4, uint32_t>,
                         (const void *)sssp_pb_expand_kernel<0, uint8_t>, (const void *)sssp_pb_expand_kernel<1, uint8_t>,
                         (const void *)sssp_pb_expand_kernel<2, uint8_t>, (const void *)sssp_pb_expand_kernel<4, uint8_t>,
                         (const void *)sssp_pb_accumulate_kernel<uint8_t>, (const void *)sssp_pb_accumulate_kernel<uint16_t>,
                         (const void *)sssp_pb_accumulate_kernel<uint32_t>};
    for (const void *fn : fns) {
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) {
        gdn_set_error("hipFuncSetAttribute(dynamic LDS): %s", hipGetErrorString(e));
        return GDN_ERR_HIP;
      }
    }
    p.dense = true;
  }
  GDN_HIP(hipDeviceSynchronize());
  p.prep_ms = t.stop_ms();
  return GDN_OK;
}

// blocking read of a small device struct through the plan's pinned block (a pageable hipMemcpy costs ~2x the latency)
// a small host struct into device memory as the ARGUMENT of a one-wave kernel: hipMemcpyAsync from pageable memory is
// staged through the runtime (a host-side copy and a DMA packet per phase), a kernel argument rides in the dispatch packet
template <typename T>
__global__ void sssp_put_kernel(T *dst, const T v) {
  const unsigned *src = reinterpret_cast<const unsigned *>(&v);
  for (unsigned i = threadIdx.x; i < sizeof(T) / 4; i += 64) reinterpret_cast<unsigned *>(dst)[i] = src[i];
}
template <typename T>
static void sssp_put(T *d_dst, const T &v) {
  static_assert(sizeof(T) % 4 == 0 && sizeof(T) <= 2048, "kernel-argument copy");
  hipLaunchKernelGGL(HIP_KERNEL_NAME(sssp_put_kernel<T>), dim3(1), dim3(64), 0, 0, d_dst, v);
}

template <typename T>
static int sssp_read(gdn_sssp_plan &p, const T *d_src, T &out) {
  return p.mail.read(const_cast<T *>(d_src), out);  // (GdnMailbox, gdn_common.hpp: no stream synchronisation per phase)
}

template <typename CT>
static void sssp_launch_sweep(gdn_sssp_plan &p, int32_t m, int32_t *d_dist) {
  const size_t lds = (sizeof(unsigned) << p.pb.log_chunk) + 16;
  CT *cand = reinterpret_cast<CT *>(p.cand.p);
  const void *W = p.w_bytes == 4 ? (const void *)p.Wp.p : (const void *)p.Wn.p;
#define SSSP_EXPAND(WB)                                                                                                      \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(sssp_pb_expand_kernel<WB, CT>), dim3(p.pb.nchunks), dim3(PB_THREADS), lds, 0, d_dist, m, \
                     p.pb.log_chunk, p.pb.chunk_ptr.p, p.pb.chunk_order.p, p.pb.U.p, p.pb.G.p, W, (unsigned)p.w_min, cand,   \
                     p.bad.p)
  switch (p.w_bytes) {
    case 0: SSSP_EXPAND(0); break;
    case 1: SSSP_EXPAND(1); break;
    case 2: SSSP_EXPAND(2); break;
    default: SSSP_EXPAND(4); break;
  }
#undef SSSP_EXPAND
  SsspTierArgs ta;
  if (p.n_tiers) {
    const unsigned n_ts = p.tier_off[p.n_tiers];
    hipLaunchKernelGGL(sssp_tier_gather_kernel, dim3(gdn_nblocks(n_ts)), dim3(GDN_BLOCK), 0, 0, d_dist, p.tier_ids.p, n_ts, p.tier_tab.p);
    ta.n = p.n_tiers;
    ta.nbins = p.pb.nbins;
    ta.rec = p.tier_rec.p;
    ta.w = p.w_bytes == 1 ? p.tier_w.p : nullptr;
    ta.ptr = p.tier_ptr.p;
    ta.cnt = p.tier_cnt.p;  // nullptr for plain streams
    ta.tab = p.tier_tab.p;
    for (int t = 0; t < p.n_tiers; t++) ta.off[t] = p.tier_off[t];
    ta.w_uniform = p.w_bytes == 0 ? (unsigned)p.w_min : 0u;
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(sssp_pb_accumulate_kernel<CT>), dim3(p.pb.nbins), dim3(PB_THREADS), lds, 0, m, p.pb.log_bin,
                     p.pb.bin_ptr.p, p.pb.bin_order.p, p.pb.V.p, cand, d_dist, p.improved.p, p.cnt.p,
                     p.bin_nbins ? p.g->rowptr : nullptr, ta);
}

static int sssp_run(gdn_sssp_plan &p, int32_t source, int32_t delta, int32_t *d_dist, gdn_stats *stats) {
  const gdn_graph *g = p.g;
  const int32_t *d_weight = p.d_weight;
  const int32_t m = g->m;
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  st.prep_ms = p.prep_ms;
  HostTimer tsolve;
  tsolve.start();  // omp_base.cc:27 t.Start()
  GDN_TRY(gdn_fill_i32(d_dist, GDN_DIST_INF, (size_t)m, 0));
  GDN_HIP(hipMemsetAsync(p.stamp.p, 0, (size_t)m * 4, 0));
  GDN_HIP(hipMemsetAsync(p.in_far.p, 0, (size_t)m * 4, 0));
  hipLaunchKernelGGL(sssp_seed_kernel, dim3(1), dim3(64), 0, 0, source, d_dist, p.near0.p);
  vid_t *near_in = p.near0.p, *near_out = p.near1.p, *far_cur = p.far0.p, *far_nxt = p.far1.p;
  unsigned n_near = 1, n_far = 0;
  unsigned long long near_edges = 0;  // out-degree sum of the NEAR list (a sweep costs ~nnz/24 worklist edges)
  int64_t thr_lo = 0, thr_hi = delta;
  int32_t pass = 0;
  int phases = 0;
  int32_t max_finite = 0;  // largest finite distance written so far
  unsigned long long relaxed_total = 0;  // edges relaxed ("x re-relaxation count"): list passes count the out-edges
                                         // of their list, a dense sweep all nnz; reported in stats.last_error
  SsspCounters h;
  ExpBigList big;
  big.items = p.bigitems.p;
  big.capacity = p.bigcap;
  const unsigned cap = p.cap;
  auto clamp = [](int64_t x) { return (int32_t)(x > GDN_DIST_INF ? GDN_DIST_INF : x); };
  const bool trace = gdn_option("GDN_SSSP_TRACE") != nullptr;  // per-phase log on stderr (tools/)
  auto wall_us = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
  double t_prev = 0;
  if (trace) {
    (void)hipDeviceSynchronize();
    t_prev = wall_us();
  }
  auto lap = [&]() { const double t = wall_us(), d = t - t_prev; t_prev = t; return d; };
  // dense sweeps start when the NEAR list owns more than nnz / dense_in out-edges and go on while more than m / dense_out
  // rows improve per sweep (tuning knobs)
  bool pre_dense_done = false;
  unsigned long long dense_pre = 192;  // = dense_in x the growth of a frontier per pass this early (GDN_SSSP_DENSE_PRE)
  if (const char *e = gdn_xoption("GDN_SSSP_DENSE_PRE")) dense_pre = atoi(e) > 0 ? (unsigned long long)atoi(e) : 0ull;
  unsigned long long dense_in = 24, dense_out = 8;  // measured on RMAT-24, U[1,255]: m/256 -> m/16 took 7.2 / 9.2 ms to 6.2 / 7.5 ms
                                                     // (round 1); with this round's worklist passes m/16 -> m/8: delta 16
                                                     // 3.9 -> 3.66 ms, unit weights 1.74 -> 1.44 ms (m/4: 4.5 / 1.46)
  if (const char *e = gdn_option("GDN_SSSP_DENSE_IN")) dense_in = atoi(e) > 0 ? (unsigned long long)atoi(e) : dense_in;
  if (const char *e = gdn_test_option("GDN_SSSP_DENSE_OUT")) dense_out = atoi(e) > 0 ? (unsigned long long)atoi(e) : dense_out;
  // light phases run inside ONE workgroup (sssp_small_kernel); GDN_SSSP_SMALL=0 keeps every phase on the host loop, =2
  // forces every phase into it that fits the lists (tests)
  unsigned small_v = SSSP_SMALL_V, small_far = SSSP_SMALL_FAR;
  if (const char *e = gdn_xoption("GDN_SSSP_SMALL_FAR")) small_far = (unsigned)atoi(e);  // FAR lists up to this long change buckets inside the one-workgroup kernel
  unsigned long long small_e = SSSP_SMALL_E;
  if (const char *e = gdn_test_option("GDN_SSSP_SMALL")) {
    if (atoi(e) == 0) small_v = 0;
    else if (atoi(e) == 2) {
      small_v = cap;
      small_e = ~0ull;
      small_far = cap;
    }
  }
  // lists beyond one workgroup go to the cooperative grid (sssp_coop_kernel) once `coop_streak` light phases in a row say
  // "high diameter" (GDN_SSSP_COOP=0: never, =1: from the first such phase -- tests)
  unsigned coop_v = 65536, coop_far = 1u << 22, coop_streak = 8, light_streak = 0;
  unsigned long long coop_e = 1ull << 20;
  if (const char *e = gdn_option("GDN_SSSP_COOP")) {
    if (atoi(e) == 0) coop_v = 0;
    else coop_streak = 0;
  }
  if (p.coop_blocks == 0) coop_v = 0;
  // Bucket width of the schedule (SsspSmallState::delta_cur): the caller's delta to begin with, doubled after a bucket whose
  // passes relaxed fewer than `light` edges (such a bucket is all launch / barrier latency), halved after one of more than
  // 8 x light, never below the caller's.  A lattice with U[1,255] weights: delta 16 as given 1.4 s and 79 K phases, 1024
  // 0.2 s (profiles/r04_sssp_delta_sweep.txt); R-MAT does not care.  GDN_SSSP_ADAPT=0: the caller's width throughout.
  bool edges_pending = false;  // near_edges of the current list is not known yet (0): the next host pass sums it
  long long delta_cur = delta;
  unsigned long long bucket_work = 0;
  unsigned light_run = 0, adapt_after = 4;
  if (const char *e = gdn_xoption("GDN_SSSP_ADAPT_AFTER")) adapt_after = (unsigned)atoi(e);
  unsigned long long light_small = 4096, light_coop = 1ull << 17, light_host = 1ull << 20;
  if (const char *e = gdn_xoption("GDN_SSSP_ADAPT")) {
    if (atoi(e) == 0) light_small = light_coop = light_host = 0;
  }
  if (const char *e = gdn_xoption("GDN_SSSP_LIGHT_SMALL")) light_small = strtoull(e, nullptr, 10);
  if (const char *e = gdn_xoption("GDN_SSSP_LIGHT_COOP")) light_coop = strtoull(e, nullptr, 10);
  if (const char *e = gdn_xoption("GDN_SSSP_LIGHT_HOST")) light_host = strtoull(e, nullptr, 10);
  for (;;) {
    if (n_near == 0 && n_far == 0) break;
    if (!pre_dense_done && (n_near > 0 || (small_v && n_far <= small_far))) {
      // (a bucket change with a short FAR list also runs inside the workgroup; after a pass that built no lists the
      // sweeps below come first)
      if (small_v && n_near <= small_v && near_edges <= small_e && (n_near > 0 || n_far <= small_far)) {
        // ---- light phases: passes and bucket changes inside one workgroup until a list outgrows it
        SsspSmallState ss;
        memset(&ss, 0, sizeof(ss));
        ss.n_near = n_near;
        ss.n_far = n_far;
        ss.near_edges = near_edges;
        ss.thr_lo = thr_lo;
        ss.thr_hi = thr_hi;
        ss.pass = pass;
        ss.near_sel = near_in == p.near1.p ? 1u : 0u;
        ss.far_sel = far_cur == p.far1.p ? 1u : 0u;
        ss.delta_cur = delta_cur;
        ss.bucket_work = bucket_work;
        ss.light = light_small;
        ss.light_run = light_run;
        ss.adapt_after = adapt_after;
        sssp_put(p.small.p, ss);
        hipLaunchKernelGGL(sssp_small_kernel, dim3(1), dim3(SSSP_SMALL_THREADS), 0, 0, g->rowptr, g->colidx, d_weight, d_dist,
                           p.stamp.p, p.in_far.p, p.near0.p, p.near1.p, p.far0.p, p.far1.p, cap, delta, small_v, small_e,
                           small_far, p.small.p);
        GDN_TRY(sssp_read(p, p.small.p, ss));
        if (ss.overflow) {
          gdn_set_error("gdn_sssp: device worklist overflow");
          return GDN_ERR_OVERFLOW;
        }
        if (trace)
          fprintf(stderr, "[sssp] %7.1f us small: %u passes, %u buckets -> status %d, bucket [%lld,%lld): near %u (%llu edges) far %u\n",
                  lap(), ss.passes, ss.buckets, ss.status, ss.thr_lo, ss.thr_hi, ss.n_near, ss.near_edges, ss.n_far);
        phases += (int)ss.passes;
        relaxed_total += ss.relaxed_edges;
        n_near = ss.n_near;
        n_far = ss.n_far;
        near_edges = ss.near_edges;
        thr_lo = ss.thr_lo;
        thr_hi = ss.thr_hi;
        pass = ss.pass;
        delta_cur = ss.delta_cur;
        bucket_work = ss.bucket_work;
        light_run = ss.light_run;
        max_finite = ss.max_dist > max_finite ? ss.max_dist : max_finite;
        near_in = ss.near_sel ? p.near1.p : p.near0.p;
        near_out = ss.near_sel ? p.near0.p : p.near1.p;
        far_cur = ss.far_sel ? p.far1.p : p.far0.p;
        far_nxt = ss.far_sel ? p.far0.p : p.far1.p;
        light_streak += ss.passes + ss.buckets;
        continue;  // status 0: both lists empty; 1: NEAR outgrew the workgroup; 2: NEAR empty, FAR too long for it
      }
    }
    if (!pre_dense_done && coop_v && light_streak >= coop_streak && n_near <= coop_v && near_edges <= coop_e && n_far <= coop_far &&
        !(p.dense && near_edges * dense_in > (unsigned long long)g->nnz)) {
      // ---- mid-size phases of a high-diameter search: passes and bucket changes on the cooperative grid
      SsspSmallState ss;
      memset(&ss, 0, sizeof(ss));
      ss.n_near = n_near;
      ss.n_far = n_far;
      ss.near_edges = near_edges;
      ss.thr_lo = thr_lo;
      ss.thr_hi = thr_hi;
      ss.pass = pass;
      ss.near_sel = near_in == p.near1.p ? 1u : 0u;
      ss.far_sel = far_cur == p.far1.p ? 1u : 0u;
      ss.delta_cur = delta_cur;
      ss.bucket_work = bucket_work;
      ss.light = light_coop;
      ss.light_run = light_run;
      ss.adapt_after = adapt_after;
      SsspCoopCnt init[3];
      memset(init, 0, sizeof(init));
      for (int k = 0; k < 3; k++) init[k].min_far = GDN_DIST_INF;
      sssp_put(p.small.p, ss);
      GDN_HIP(hipMemcpyAsync(p.coop_cnt.p, init, sizeof(init), hipMemcpyHostToDevice, 0));
      GDN_HIP(hipMemsetAsync(p.coop_bar.p, 0, GDN_GBAR_WORDS * sizeof(unsigned), 0));
      const eoff_t *a_rowptr = g->rowptr;
      const vid_t *a_colidx = g->colidx;
      const int32_t *a_w = d_weight;
      int32_t *a_dist = d_dist, *a_stamp = p.stamp.p;
      unsigned *a_in_far = p.in_far.p;
      vid_t *a_n0 = p.near0.p, *a_n1 = p.near1.p, *a_f0 = p.far0.p, *a_f1 = p.far1.p;
      unsigned a_cap = cap, a_max_v = coop_v, a_max_far = coop_far;
      int32_t a_delta = delta;
      unsigned long long a_max_e = coop_e;
      SsspCoopCnt *a_cnt = p.coop_cnt.p;
      unsigned *a_bar = p.coop_bar.p;
      SsspSmallState *a_state = p.small.p;
      void *args[] = {&a_rowptr, &a_colidx, &a_w, &a_dist, &a_stamp, &a_in_far, &a_n0, &a_n1, &a_f0, &a_f1, &a_cap, &a_delta,
                      &a_max_v, &a_max_e, &a_max_far, &a_cnt, &a_bar, &a_state};
      GDN_HIP(hipLaunchCooperativeKernel((const void *)sssp_coop_kernel, dim3((unsigned)p.coop_blocks), dim3(SSSP_COOP_THREADS), args, 0, 0));
      GDN_TRY(sssp_read(p, p.small.p, ss));
      if (ss.overflow) {
        gdn_set_error("gdn_sssp: device worklist overflow");
        return GDN_ERR_OVERFLOW;
      }
      if (trace)
        fprintf(stderr, "[sssp] %7.1f us coop: %u passes, %u buckets -> status %d, bucket [%lld,%lld): near %u (%llu edges) far %u\n",
                lap(), ss.passes, ss.buckets, ss.status, ss.thr_lo, ss.thr_hi, ss.n_near, ss.near_edges, ss.n_far);
      phases += (int)ss.passes;
      relaxed_total += ss.relaxed_edges;
      n_near = ss.n_near;
      n_far = ss.n_far;
      near_edges = ss.near_edges;
      thr_lo = ss.thr_lo;
      thr_hi = ss.thr_hi;
      pass = ss.pass;
      delta_cur = ss.delta_cur;
      bucket_work = ss.bucket_work;
      light_run = ss.light_run;
      max_finite = ss.max_dist > max_finite ? ss.max_dist : max_finite;
      near_in = ss.near_sel ? p.near1.p : p.near0.p;
      near_out = ss.near_sel ? p.near0.p : p.near1.p;
      far_cur = ss.far_sel ? p.far1.p : p.far0.p;
      far_nxt = ss.far_sel ? p.far0.p : p.far1.p;
      light_streak += ss.passes + ss.buckets;
      if (ss.passes + ss.buckets == 0) coop_v = 0;  // (cannot happen: the entry test mirrors the kernel's; no endless loop)
      continue;
    }
    if (n_near > 0) {
      if (p.dense && (near_edges * dense_in > (unsigned long long)g->nnz || pre_dense_done)) {
        pre_dense_done = false;
        // ---- heavy frontier: Bellman-Ford sweeps over all edges until few rows still improve
        light_streak = 0;
        unsigned long long improved = 0, imp_edges = 0;
        // Between the sweeps and the worklist tail: BINNED passes (sssp_bin_*) while the rows improved by the last step
        // own at most nnz / SSSP_BIN_FRAC out-edges -- they relax those edges only, still without a random access per edge.
        // They go on while the improved rows own more than nnz / bin_out edges (GDN_SSSP_BIN_OUT), the sweeps while more
        // than m / dense_out rows improve.
        const bool use_bins = p.bin_nbins > 0;
        unsigned long long bin_out = 64;
        if (const char *e = gdn_test_option("GDN_SSSP_BIN_OUT")) bin_out = atoi(e) > 0 ? (unsigned long long)atoi(e) : bin_out;
        bool have_queue = false;  // near_in holds the rows improved by the last step (n_q of them)
        unsigned n_q = 0;
        bool more;
        do {
          ++phases;
          memset(&h, 0, sizeof(h));
          h.min_far = GDN_DIST_INF;
          sssp_put(p.cnt.p, h);
          bool binned = use_bins && have_queue && n_q > 0 && imp_edges * SSSP_BIN_FRAC <= (unsigned long long)g->nnz;
          int cbits = 0;
          if (binned) {
            SsspBinVis bv;
            bv.colidx = g->colidx;
            bv.weight = d_weight;
            bv.dist = d_dist;
            bv.buf = p.bin_buf.p;
            bv.cur = p.bin_cur.p;
            bv.overflow = &p.cnt.p->overflow;
            bv.cap_each = p.bin_cap_each;
            bv.sub = 0;
            bv.bin_bits = p.bin_bits;
            bv.du = 0;
            big.count = &p.cnt.p->big_count;
            big.overflow = &p.cnt.p->overflow;
            big.min_deg = (unsigned)EXP_BIG;
            hipLaunchKernelGGL(sssp_bin_kernel, dim3(gdn_nblocks(n_q)), dim3(GDN_BLOCK), 0, 0, g->rowptr, near_in, n_q, big, bv);
            hipLaunchKernelGGL(sssp_bin_big_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, bv);
            hipLaunchKernelGGL(sssp_bin_apply_kernel, dim3(p.bin_nbins), dim3(SSSP_BIN_THREADS), (size_t)4 << SSSP_BIN_LOGB, 0,
                               p.bin_buf.p, p.bin_cur.p, &p.cnt.p->overflow, p.bin_cap_each, m, d_dist, p.improved.p, g->rowptr,
                               p.cnt.p, near_out, cap);
            GDN_TRY(sssp_read(p, p.cnt.p, h));
            if (h.overflow) {  // a list was too short (or the item list): nothing was applied, the step is a sweep instead
              if (trace) fprintf(stderr, "[sssp] %7.1f us phase %d binned pass overflowed its lists: repeated as a sweep\n", lap(), phases);
              binned = false;
              memset(&h, 0, sizeof(h));
              h.min_far = GDN_DIST_INF;
              sssp_put(p.cnt.p, h);
            } else {
              relaxed_total += imp_edges;
              vid_t *t = near_in;
              near_in = near_out;
              near_out = t;
              n_q = h.near_count;
              have_queue = true;
            }
          }
          if (!binned) {
            relaxed_total += g->nnz;
            // 8- / 16-bit candidates while every finite candidate (a finite distance + a weight) stays below 0xFF / 0xFFFF
            // (GDN_SSSP_CAND32 / GDN_SSSP_CAND16: test knobs that keep the wider form)
            const int64_t bound = (int64_t)max_finite + (int64_t)p.w_max;
            cbits = gdn_test_option("GDN_SSSP_CAND32") ? 32 : (bound < 0xFF && !gdn_test_option("GDN_SSSP_CAND16")) ? 8 : bound < 0xFFFF ? 16 : 32;
            // (measured and dropped: Gauss-Seidel sweeps -- expand + accumulate per quarter of the bins, so that rows improved
            // in an earlier quarter are sources again inside the same sweep -- need 4 sweeps instead of 5 on RMAT-24 U[1,255],
            // but each costs 1.05 ms instead of 0.61: every partial launch reloads the whole distance slice)
            if (p.cand_bits != 0 && p.cand_bits != cbits)  // (found by the randomised sweep: rows 0 of bins took stale 8-bit
                                                           // candidates read as 16-bit ones, profiles/sessions/r04_67.sh)
              GDN_TRY(gdn_fill_i32(reinterpret_cast<int32_t *>(p.cand.p), -1, (size_t)p.pb.n_pad + 8, 0));
            p.cand_bits = cbits;
            if (cbits == 8) sssp_launch_sweep<uint8_t>(p, m, d_dist);
            else if (cbits == 16) sssp_launch_sweep<uint16_t>(p, m, d_dist);
            else sssp_launch_sweep<uint32_t>(p, m, d_dist);
            GDN_TRY(sssp_read(p, p.cnt.p, h));
            have_queue = false;
          }
          improved = h.relaxed;
          imp_edges = h.improved_edges;
          max_finite = h.max_dist > max_finite ? h.max_dist : max_finite;
          if (trace) {
            const double us = lap();
            if (binned)
              fprintf(stderr, "[sssp] %7.1f us phase %d binned pass: %llu rows improved (%llu out-edges = %.1f %% of the graph), max distance %d\n",
                      us, phases, improved, imp_edges, 100.0 * (double)imp_edges / (double)(g->nnz ? g->nnz : 1), max_finite);
            else {
              unsigned long long ie = imp_edges;
              if (!use_bins) {  // (trace only) the out-edges of the improved rows
                DevBuf<unsigned long long> d_ie;
                if (d_ie.alloc(1) == GDN_OK && hipMemset(d_ie.p, 0, 8) == hipSuccess) {
                  hipLaunchKernelGGL(sssp_bitmap_edges_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, p.improved.p, p.nwords, m, g->rowptr, d_ie.p);
                  (void)hipMemcpy(&ie, d_ie.p, 8, hipMemcpyDeviceToHost);
                }
              }
              fprintf(stderr, "[sssp] %7.1f us phase %d dense sweep (%d-byte weights, %d-bit candidates): %llu rows improved (%llu out-edges = %.1f %% of the graph), max distance %d\n",
                      us, phases, p.w_bytes, cbits, improved, ie, 100.0 * (double)ie / (double)(g->nnz ? g->nnz : 1), max_finite);
            }
            (void)lap();
          }
          more = improved * dense_out > (unsigned long long)m ||
                 (use_bins && improved > 0 && imp_edges * bin_out > (unsigned long long)g->nnz && imp_edges * SSSP_BIN_FRAC <= (unsigned long long)g->nnz);
          if (more && use_bins && !have_queue && imp_edges * SSSP_BIN_FRAC <= (unsigned long long)g->nnz) {
            // the next step is a binned pass: it needs the improved rows as a list
            memset(&h, 0, sizeof(h));
            h.min_far = GDN_DIST_INF;
            sssp_put(p.cnt.p, h);
            hipLaunchKernelGGL(sssp_bitmap_to_queue, dim3(SSSP_B2Q_BLOCKS), dim3(GDN_BLOCK), 0, 0, p.improved.p, p.nwords, m, near_in,
                               p.cnt.p, cap, g->rowptr);
            GDN_TRY(sssp_read(p, p.cnt.p, h));
            n_q = h.near_count;
            have_queue = true;
            if (trace) fprintf(stderr, "[sssp] %7.1f us improved rows -> list of %u (%llu edges) for a binned pass\n", lap(), n_q, h.relaxed);
          }
        } while (more);
        // the rows improved by the LAST step are the only ones with unpropagated distances: they
        // become a plain Bellman-Ford worklist (one infinite bucket); the parked FAR list is
        // covered by the sweeps and dropped
        GDN_HIP(hipMemsetAsync(p.in_far.p, 0, (size_t)m * 4, 0));
        if (have_queue) {  // the last binned pass left them in near_in already
          h.near_count = n_q;
          h.relaxed = imp_edges;
          if (trace) fprintf(stderr, "[sssp] %7.1f us worklist tail from the list of %u (%llu edges)\n", lap(), n_q, imp_edges);
        } else {
          // a long list: its out-degree sum (two gathers per row: most of this conversion's time) is left to the first tail
          // pass, which reads the row offsets anyway (edges_pending); nothing below decides anything on it for such a list
          const bool lazy_edges = improved > 65536ull;
          memset(&h, 0, sizeof(h));
          h.min_far = GDN_DIST_INF;
          sssp_put(p.cnt.p, h);
          hipLaunchKernelGGL(sssp_bitmap_to_queue, dim3(SSSP_B2Q_BLOCKS), dim3(GDN_BLOCK), 0, 0, p.improved.p,
                             p.nwords, m, near_in, p.cnt.p, cap, lazy_edges ? (const eoff_t *)nullptr : g->rowptr);
          GDN_TRY(sssp_read(p, p.cnt.p, h));
          edges_pending = lazy_edges && h.near_count > 65536u;
          if (lazy_edges && !edges_pending) {  // (cannot happen: the list holds exactly the improved rows)
            gdn_set_error("gdn_sssp: the improved rows and their list disagree (internal error)");
            return GDN_ERR_INVALID;
          }
          if (trace) fprintf(stderr, "[sssp] %7.1f us improved rows -> queue of %u (%llu edges%s)\n", lap(), h.near_count, h.relaxed,
                             edges_pending ? ": summed by the first pass" : "");
        }
        n_near = h.near_count;
        n_far = 0;
        near_edges = h.relaxed;  // out-degree sum of the queue (counted by the conversion)
        thr_lo = 0;
        thr_hi = (int64_t)GDN_DIST_INF;
        if (near_edges * dense_in > (unsigned long long)g->nnz) near_edges = (unsigned long long)g->nnz / dense_in;  // no way back into the sweeps
        continue;
      }
      // A pass whose input already owns more than nnz / dense_pre out-edges will be followed by the dense sweeps on a
      // low-diameter graph (the frontier grows several-fold per pass): it then only lowers distances -- no stamp, no FAR
      // flag, no degree read, no list -- since the sweeps work from the distances alone and drop both lists anyway.
      const bool pre_dense = p.dense && thr_hi < (int64_t)GDN_DIST_INF && near_edges * dense_pre > (unsigned long long)g->nnz;
      ++pass;
      ++phases;
      relaxed_total += near_edges;
      bucket_work += near_edges;
      h.near_count = 0;
      h.far_count = n_far;
      h.big_count = 0;
      h.overflow = 0;
      h.min_far = GDN_DIST_INF;
      h.max_dist = 0;
      h.relaxed = 0;
      h.improved_edges = 0;
      sssp_put(p.cnt.p, h);
      SsspVis vis;
      vis.rowptr = g->rowptr;
      vis.near_edges = 0;
      vis.colidx = g->colidx;
      vis.weight = d_weight;
      vis.dist = d_dist;
      vis.stamp = p.stamp.p;
      vis.in_far = p.in_far.p;
      vis.near_out = near_out;
      vis.far_out = far_cur;  // FAR grows in place behind its current tail
      vis.cnt = p.cnt.p;
      vis.cap = cap;
      vis.thr_hi = clamp(thr_hi);
      vis.pass = pass;
      vis.no_push = pre_dense ? 1 : 0;
      vis.du = 0;
      vis.max_d = 0;
      big.count = &p.cnt.p->big_count;
      big.overflow = &p.cnt.p->overflow;
      // a short list of long rows: hand every row of a wave's width or more to the persistent item kernel (walked one
      // after the other by the few waves of such a pass they cost 0.4 ms on RMAT-24; gdn_bfs.hip does the same)
      big.min_deg = ((uint64_t)n_near < 65536u && (uint64_t)n_near + near_edges / EXP_CHUNK + 1024u < (uint64_t)p.bigcap)
                        ? 64u : (unsigned)EXP_BIG;  // (on a 228 K-vertex list the item detour cost 0.42 ms instead of 0.28)
      if (const char *e = gdn_xoption("GDN_SSSP_MINDEG")) big.min_deg = (unsigned)atoi(e);
      hipLaunchKernelGGL(sssp_relax_kernel, dim3(gdn_nblocks(n_near) < SSSP_RELAX_GRID ? gdn_nblocks(n_near) : SSSP_RELAX_GRID),
                         dim3(GDN_BLOCK), 0, 0, g->rowptr, near_in,
                         n_near, clamp(thr_lo), big, vis, edges_pending ? &p.cnt.p->improved_edges : (unsigned long long *)nullptr);
      hipLaunchKernelGGL(sssp_relax_big_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
      GDN_TRY(sssp_read(p, p.cnt.p, h));
      if (h.overflow) {
        gdn_set_error("gdn_sssp: device worklist overflow");
        return GDN_ERR_OVERFLOW;
      }
      if (trace)
        fprintf(stderr, "[sssp] %7.1f us phase %d relax [%lld,%lld): near %u (%llu edges) -> near %u (%llu edges) far %u big %u\n", lap(), phases,
                (long long)thr_lo, (long long)thr_hi, n_near, near_edges, h.near_count, h.relaxed, h.far_count, h.big_count);
      max_finite = h.max_dist > max_finite ? h.max_dist : max_finite;
      light_streak++;
      if (edges_pending) {  // the out-edges of the list this pass walked (see the conversion behind the sweeps)
        relaxed_total += h.improved_edges;
        bucket_work += h.improved_edges;  // (sssp_adapt_delta must see the pass's real work, not the 0 it was queued with)
        edges_pending = false;
      }
      if (pre_dense) {  // no lists were built: the sweeps take over from the distances
        pre_dense_done = true;
        n_near = 1;  // (placeholder: the dense branch does not read the list)
        continue;
      }
      n_near = h.near_count;
      n_far = h.far_count;
      near_edges = h.relaxed;
      vid_t *t = near_in;
      near_in = near_out;
      near_out = t;
      continue;
    }
    // ---- next non-empty bucket (omp_base.cc:66-72 votes for the smallest non-empty bin).  As inside the one-workgroup and
    // cooperative kernels: the split is run at once for the bucket right BEHIND the old one and collects the minimum of what
    // it keeps; only when nothing moved does a second pass split at the bucket of that minimum -- one launch and one read
    // back per bucket change where the minimum-first order took two of each.
    delta_cur = sssp_adapt_delta(delta_cur, (long long)delta, bucket_work, light_host, light_run, adapt_after);
    bucket_work = 0;
    int64_t spec_lo = thr_hi, spec_hi = thr_hi + delta_cur;
    bool finished = false;
    for (;;) {
      h.near_count = 0;
      h.far_count = 0;
      h.big_count = 0;
      h.overflow = 0;
      h.min_far = GDN_DIST_INF;
      h.max_dist = 0;
      h.relaxed = 0;
      sssp_put(p.cnt.p, h);
      hipLaunchKernelGGL(sssp_far_split_kernel, dim3(gdn_nblocks(n_far) < 2048u ? gdn_nblocks(n_far) : 2048u), dim3(GDN_BLOCK), 0, 0,
                         far_cur, n_far, d_dist,
                         clamp(thr_hi), clamp(spec_hi), p.in_far.p, near_in, far_nxt, p.cnt.p, cap, g->rowptr);
      GDN_TRY(sssp_read(p, p.cnt.p, h));
      if (h.overflow) {
        gdn_set_error("gdn_sssp: device worklist overflow");
        return GDN_ERR_OVERFLOW;
      }
      if (trace)
        fprintf(stderr, "[sssp] %7.1f us split far %u for bucket [%lld,%lld): near %u (%llu edges) far %u (min %d)\n", lap(), n_far,
                (long long)spec_lo, (long long)spec_hi, h.near_count, h.relaxed, h.far_count, h.min_far);
      n_far = h.far_count;
      vid_t *t = far_cur;
      far_cur = far_nxt;
      far_nxt = t;
      if (h.near_count > 0 || n_far == 0) {
        if (h.near_count > 0) {
          thr_lo = spec_lo;
          thr_hi = spec_hi;
        }
        n_near = h.near_count;
        near_edges = h.relaxed;
        finished = n_near == 0;  // only stale entries were left
        break;
      }
      thr_hi = spec_hi;  // that bucket was empty and everything kept lies behind it: nothing is stale against it
      spec_lo = ((int64_t)h.min_far / delta_cur) * (int64_t)delta_cur;
      if (spec_lo < thr_hi) spec_lo = thr_hi;
      spec_hi = spec_lo + delta_cur;
    }
    if (finished) break;
  }
  GDN_HIP(hipGetLastError());
  if (p.dense) {
    unsigned bad = 0;
    GDN_HIP(hipMemcpy(&bad, p.bad.p, 4, hipMemcpyDeviceToHost));
    if (bad) {
      gdn_set_error("gdn_sssp: a 16-bit candidate distance overflowed (internal error)");
      return GDN_ERR_OVERFLOW;
    }
  }
  st.solve_ms = tsolve.stop_ms();
  st.iterations = phases;
  uint64_t te = 0;
  GDN_TRY(gdn_reached_edges(g, d_dist, GDN_DIST_INF, &te));
  st.edges_traversed = te;
  st.last_error = (double)relaxed_total;  // SSSP: edges relaxed over the whole solve (exact below 2^53)
  if (stats) *stats = st;
  return GDN_OK;
}

extern "C" {

int gdn_sssp_plan_create(const gdn_graph *g, const int32_t *d_weight, int32_t dense, gdn_sssp_plan **plan) {
  GDN_REQUIRE(plan != nullptr, "plan");
  *plan = nullptr;
  GDN_REQUIRE(g != nullptr && (d_weight != nullptr || g->nnz == 0), "graph / d_weight");
  gdn_sssp_plan *p = new gdn_sssp_plan();
  // equal weights >= 1 (from 2^22 edges on; GDN_SSSP_UNIT_BFS=0 never, =1 at any size): the BFS route -- the blocked layout
  // of the sweeps is then not built at all
  bool bfs_route = false;
  int32_t w_all = 0;
  {
    const char *e = gdn_option("GDN_SSSP_UNIT_BFS");
    const bool want = dense != 0 && g->nnz > 0 && !(e && e[0] == '0') && (g->nnz >= (1ull << 22) || (e && e[0] == '1'));
    if (want) {
      DevBuf<int32_t> rng;
      if (rng.alloc(2) == GDN_OK) {
        const int32_t init[2] = {0x7FFFFFFF, -0x7FFFFFFF - 1};
        int32_t h[2] = {0, 1};
        if (hipMemcpy(rng.p, init, 8, hipMemcpyHostToDevice) == hipSuccess) {
          hipLaunchKernelGGL(sssp_weight_range_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, d_weight, (size_t)g->nnz, rng.p);
          if (hipMemcpy(h, rng.p, 8, hipMemcpyDeviceToHost) == hipSuccess && h[0] == h[1] && h[0] >= 1) {
            bfs_route = true;
            w_all = h[0];
          }
        }
      }
    }
  }
  int rc = sssp_plan_init(*p, g, d_weight, dense != 0 && !bfs_route);
  if (rc == GDN_OK && bfs_route) {
    HostTimer tb;
    tb.start();
    int rb = gdn_graph_transpose(g, &p->bfs_gin);
    if (rb == GDN_OK) rb = gdn_bfs_plan_create(g, p->bfs_gin, 1, &p->bfs);
    if (rb == GDN_OK) {
      p->bfs_w = w_all;
      p->prep_ms += tb.stop_ms();
    } else {  // no room for the transpose / the search plan: the sweeps after all
      delete p;
      gdn_scratch_trim();
      p = new gdn_sssp_plan();
      rc = sssp_plan_init(*p, g, d_weight, dense != 0);
    }
  }
  if (rc != GDN_OK) {
    delete p;
    return rc;
  }
  *plan = p;
  return GDN_OK;
}

int gdn_sssp_plan_free(gdn_sssp_plan *plan) {
  delete plan;
  return GDN_OK;
}

int gdn_sssp_run(gdn_sssp_plan *plan, int32_t source, int32_t delta, int32_t *d_dist, gdn_stats *stats) {
  GDN_REQUIRE(plan != nullptr && d_dist != nullptr, "plan / d_dist");
  GDN_REQUIRE(source >= 0 && source < plan->g->m, "source out of range");
  GDN_REQUIRE(delta >= 1, "delta must be >= 1");
  if (plan->bfs) {  // equal weights: hop counts x weight (any delta gives these distances)
    gdn_stats bs;
    memset(&bs, 0, sizeof(bs));
    GDN_TRY(gdn_bfs_run(plan->bfs, source, d_dist, &bs));  // (solve_ms: the search's own timed region)
    HostTimer t;
    t.start();
    hipLaunchKernelGGL(sssp_depth_to_dist_kernel, dim3(gdn_nblocks((uint64_t)plan->g->m)), dim3(GDN_BLOCK), 0, 0, d_dist, plan->g->m,
                       plan->bfs_w);
    GDN_HIP(hipDeviceSynchronize());
    bs.solve_ms += t.stop_ms();
    bs.prep_ms = plan->prep_ms;
    bs.last_error = (double)bs.edges_traversed;  // SSSP: edges relaxed -- every out-edge of a reached vertex once
    bs.reserved = 1;                             // (include/gardenia_hip.h: 1 = solved through the BFS plan)
    if (stats) *stats = bs;
    return GDN_OK;
  }
  return sssp_run(*plan, source, delta, d_dist, stats);
}

int gdn_sssp_dev(const gdn_graph *g, const int32_t *d_weight, int32_t source, int32_t delta, int32_t *d_dist,
                 gdn_stats *stats) {
  GDN_REQUIRE(g != nullptr && d_dist != nullptr && (d_weight != nullptr || g->nnz == 0), "null argument");
  GDN_REQUIRE(source >= 0 && source < g->m, "source out of range");
  GDN_REQUIRE(delta >= 1, "delta must be >= 1");
  // The blocked layout the dense sweeps need is built INSIDE the call from 2^24 edges on and reported as prep_ms -- where
  // the reference's blocked solvers do their preprocessing (before t.Start(): src/pr/push_pb.cu:271,339,
  // include/segmenting.h:31-176).  RMAT-24: 33 ms of build against a solve that drops from 32.6 to 3.6 ms; below 2^24 edges
  // the worklists alone are faster than the build.  Not cached across calls: a caller that solves from many sources
  // holds a plan (gdn_sssp_plan_create / gdn_sssp_run) -- keying a hidden cache on caller pointers would return stale
  // layouts for arrays rewritten in place.  GDN_SSSP_ONESHOT_DENSE_MIN moves the threshold (0 = never).
  unsigned long long dense_min = 1ull << 24;
  if (const char *e = gdn_test_option("GDN_SSSP_ONESHOT_DENSE_MIN")) dense_min = strtoull(e, nullptr, 10);
  const bool want_dense = dense_min != 0 && g->nnz >= dense_min;
  {
    gdn_sssp_plan p;
    // (no lists for the binned passes here: allocating them costs more than they save in ONE solve)
    const int rc = sssp_plan_init(p, g, d_weight, want_dense, /*bins=*/false);
    if (rc == GDN_OK) return sssp_run(p, source, delta, d_dist, stats);
    if (!(rc == GDN_ERR_OOM && want_dense)) return rc;
  }
  // the blocked layout (~10 B per edge + its build scratch) did not fit: the worklist-only plan needs none of it
  // (the first plan and everything it held is released by now)
  gdn_scratch_trim();
  gdn_sssp_plan q;
  GDN_TRY(sssp_plan_init(q, g, d_weight, /*dense=*/false, /*bins=*/false));
  return sssp_run(q, source, delta, d_dist, stats);
}

// Host API: one call == SSSPSolver(g, source, weight, dist, delta) (src/sssp/main.cc:27).
int gdn_sssp(int32_t m, uint64_t nnz, const uint64_t *rowptr, const int32_t *colidx, const int32_t *weight,
             int32_t source, int32_t delta, int32_t *dist, gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && rowptr && dist && (weight || nnz == 0), "null argument");
  GDN_REQUIRE(source >= 0 && source < m, "source out of range");
  GDN_REQUIRE(delta >= 1, "delta must be >= 1");
  GDN_TRY(gdn_require_device());
  HostTimer th2d;
  th2d.start();
  gdn_graph *g = nullptr;
  GDN_TRY(gdn_graph_upload(m, nnz, rowptr, colidx, &g));
  DevBuf<int32_t> d_w, d_dist;
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  int rc = GDN_OK;
  do {
    if ((rc = d_w.alloc(nnz)) || (rc = d_dist.alloc(m))) break;
    if (nnz && hipMemcpy(d_w.p, weight, nnz * 4, hipMemcpyHostToDevice) != hipSuccess) {
      gdn_set_error("gdn_sssp: upload failed");
      rc = GDN_ERR_HIP;
      break;
    }
    const double h2d = th2d.stop_ms();
    if ((rc = gdn_sssp_dev(g, d_w.p, source, delta, d_dist.p, &st))) break;
    st.h2d_ms = h2d;
    if (hipMemcpy(dist, d_dist.p, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_sssp: download failed");
      rc = GDN_ERR_HIP;
    }
  } while (0);
  gdn_graph_free(g);
  if (stats) *stats = st;
  return rc;
}

}  // extern "C"
