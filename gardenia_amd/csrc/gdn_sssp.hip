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

#include "gdn_expand.hpp"
#include "gdn_pb.hpp"

struct SsspCounters {
  unsigned near_count;
  unsigned far_count;
  unsigned big_count;
  unsigned overflow;
  int min_far;
  unsigned pad;
  unsigned long long relaxed;
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
  int32_t du;  // per-lane: distance of this lane's source vertex
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
          if (nd < thr_hi) {
            push_near = atomicExch(&stamp[dst], pass) != pass;
            if (push_near) near_edges += rowptr[dst + 1] - rowptr[dst];
          } else {
            push_far = atomicExch(&in_far[dst], 1u) == 0u;
          }
        }
      }
    }
    gdn_wl_push_staged(near_st, near_out, &cnt->near_count, cap, push_near, dst, &cnt->overflow);
    gdn_wl_push_staged(far_st, far_out, &cnt->far_count, cap, push_far, dst, &cnt->overflow);
  }
  __device__ __forceinline__ void finish() {
    gdn_wl_flush(near_st, near_out, &cnt->near_count, cap, &cnt->overflow);
    gdn_wl_flush(far_st, far_out, &cnt->far_count, cap, &cnt->overflow);
    const unsigned long long s = gdn_wave_sum(near_edges);
    if (gdn_lane() == 0 && s) atomicAdd(&cnt->relaxed, s);
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
sssp_relax_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ near_in, unsigned n,
                  int32_t thr_lo, ExpBigList big, SsspVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  __shared__ vid_t s_stage[2][GDN_WAVES_PER_BLOCK][GDN_WL_STAGE];
  vis.near_st.strip = s_stage[0][threadIdx.x >> 6];
  vis.far_st.strip = s_stage[1][threadIdx.x >> 6];
  vis.near_st.n = vis.far_st.n = 0;
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
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
    }
  }
  vis.near_edges = 0;
  gdn_expand_wave(b, e, v, big, vis, s_scan[threadIdx.x >> 6]);
  vis.finish();
}

__global__ void __launch_bounds__(GDN_BLOCK)
sssp_relax_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, SsspVis vis) {
  __shared__ vid_t s_stage[2][GDN_WAVES_PER_BLOCK][GDN_WL_STAGE];
  vis.near_st.strip = s_stage[0][threadIdx.x >> 6];
  vis.far_st.strip = s_stage[1][threadIdx.x >> 6];
  vis.near_st.n = vis.far_st.n = 0;
  vis.du = 0;
  vis.near_edges = 0;
  gdn_expand_big_items(rowptr, big, vis);
  vis.finish();
}

// smallest distance parked in FAR that is still >= thr_hi (stale entries are ignored)
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_far_min_kernel(const vid_t *__restrict__ far_in, unsigned n, const int32_t *__restrict__ dist,
                    int32_t thr_hi, SsspCounters *cnt) {
  // persistent grid, ONE atomic per workgroup (the counter is a single hot address)
  __shared__ int32_t s_min[GDN_WAVES_PER_BLOCK];
  int32_t d = GDN_DIST_INF;
  for (unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x; i < n; i += gridDim.x * GDN_BLOCK) {
    const int32_t x = dist[far_in[i]];
    if (x >= thr_hi && x < d) d = x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int32_t t = __shfl_xor(d, o, 64);
    d = t < d ? t : d;
  }
  if (gdn_lane() == 0) s_min[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < GDN_WAVES_PER_BLOCK; w++) d = s_min[w] < d ? s_min[w] : d;
    if (d != GDN_DIST_INF) atomicMin(&cnt->min_far, d);
  }
}

// FAR -> {NEAR of the new bucket, FAR kept, dropped}
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
  const unsigned stride = gridDim.x * GDN_BLOCK;
  for (unsigned i0 = blockIdx.x * GDN_BLOCK; i0 < n; i0 += stride) {  // wave-uniform trip count
    const unsigned i = i0 + threadIdx.x;
    bool to_near = false, to_far = false;
    vid_t w = 0;
    if (i < n) {
      w = far_in[i];
      const int32_t d = dist[w];
      if (d >= new_hi) to_far = true;
      else {
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
typedef unsigned short sssp_u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned sssp_u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(PB_THREADS)
sssp_pb_expand_kernel(const int32_t *__restrict__ dist, int32_t m_src, int log_chunk,
                      const eoff_t *__restrict__ chunk_ptr, const uint32_t *__restrict__ chunk_order,
                      const uint16_t *__restrict__ U, const uint32_t *__restrict__ G,
                      const uint32_t *__restrict__ W, unsigned *__restrict__ cand) {
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
  const eoff_t h0 = chunk_ptr[c] >> 2, h1 = chunk_ptr[c + 1] >> 2;
  const sssp_u16x4 *U4 = reinterpret_cast<const sssp_u16x4 *>(U);
  const sssp_u32x4 *W4 = reinterpret_cast<const sssp_u32x4 *>(W);
  sssp_u32x4 *C4 = reinterpret_cast<sssp_u32x4 *>(cand);
  constexpr int UNR = 4;
  for (eoff_t h = h0 + threadIdx.x; h < h1; h += UNR * PB_THREADS) {
    sssp_u16x4 u[UNR];
    sssp_u32x4 w[UNR];
    unsigned d[UNR];
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t hh = h + (eoff_t)r * PB_THREADS;
      if (hh < h1) {
        u[r] = __builtin_nontemporal_load(U4 + hh);
        w[r] = __builtin_nontemporal_load(W4 + hh);
        d[r] = __builtin_nontemporal_load(G + (hh >> 1));
      }
    }
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t hh = h + (eoff_t)r * PB_THREADS;
      if (hh < h1) {
        const unsigned INF = (unsigned)GDN_DIST_INF;
        sssp_u32x4 o;
        unsigned t;
        t = s_d[u[r].x]; t = t >= INF ? INF : t + w[r].x; o.x = t < INF ? t : INF;
        t = s_d[u[r].y]; t = t >= INF ? INF : t + w[r].y; o.y = t < INF ? t : INF;
        t = s_d[u[r].z]; t = t >= INF ? INF : t + w[r].z; o.z = t < INF ? t : INF;
        t = s_d[u[r].w]; t = t >= INF ? INF : t + w[r].w; o.w = t < INF ? t : INF;
        C4[2 * (size_t)d[r] + (size_t)(hh & 1)] = o;
      }
    }
  }
}

__global__ void __launch_bounds__(PB_THREADS)
sssp_pb_accumulate_kernel(int32_t m_dst, int log_bin, const eoff_t *__restrict__ bin_ptr,
                          const uint32_t *__restrict__ bin_order, const uint16_t *__restrict__ V,
                          const unsigned *__restrict__ cand, int32_t *__restrict__ dist,
                          unsigned *__restrict__ improved_bits, SsspCounters *cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_min[];
  __shared__ unsigned long long s_red[PB_WAVES];
  const unsigned bn = 1u << log_bin;
  const unsigned b = bin_order[blockIdx.x];
  for (unsigned i = threadIdx.x; i < bn; i += PB_THREADS) s_min[i] = (unsigned)GDN_DIST_INF;
  __syncthreads();
  const eoff_t q0 = bin_ptr[b] >> 2, q1 = bin_ptr[b + 1] >> 2;
  const sssp_u32x4 *C4 = reinterpret_cast<const sssp_u32x4 *>(cand);
  const sssp_u16x4 *V4 = reinterpret_cast<const sssp_u16x4 *>(V);
  constexpr int UNR = 4;
  for (eoff_t q = q0 + threadIdx.x; q < q1; q += UNR * PB_THREADS) {
    sssp_u32x4 xs[UNR];
    sssp_u16x4 vs[UNR];
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t qq = q + (eoff_t)r * PB_THREADS;
      if (qq < q1) {
        xs[r] = __builtin_nontemporal_load(C4 + qq);
        vs[r] = __builtin_nontemporal_load(V4 + qq);
      }
    }
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t qq = q + (eoff_t)r * PB_THREADS;
      if (qq < q1) {
        const unsigned INF = (unsigned)GDN_DIST_INF;
        if (xs[r].x < INF) atomicMin(&s_min[vs[r].x], xs[r].x);
        if (xs[r].y < INF) atomicMin(&s_min[vs[r].y], xs[r].y);
        if (xs[r].z < INF) atomicMin(&s_min[vs[r].z], xs[r].z);
        if (xs[r].w < INF) atomicMin(&s_min[vs[r].w], xs[r].w);
      }
    }
  }
  __syncthreads();
  // epilogue: one row per thread and step; a wave covers 64 consecutive rows = 2 bitmap words
  const unsigned lane = gdn_lane();
  const size_t row0 = (size_t)b << log_bin;
  unsigned long long improved = 0;
  for (unsigned i = threadIdx.x; i < bn; i += PB_THREADS) {
    const size_t row = row0 + i;
    bool imp = false;
    if (row < (size_t)m_dst) {
      const unsigned nm = s_min[i];
      const unsigned old = (unsigned)dist[row];
      if (nm < old) {
        dist[row] = (int32_t)nm;
        imp = true;
      }
    }
    const unsigned long long mask = __ballot(imp);
    if ((lane & 31u) == 0) improved_bits[(row0 + i) >> 5] = (unsigned)(mask >> (lane & 32u));
    if (lane == 0) improved += (unsigned long long)__popcll(mask);
  }
  improved = gdn_wave_sum(improved);
  if (lane == 0) s_red[threadIdx.x >> 6] = improved;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < PB_WAVES; i++) t += s_red[i];
    if (t) atomicAdd(&cnt->relaxed, t);
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
sssp_bitmap_to_queue(const unsigned *__restrict__ bits, unsigned nwords, int32_t m, vid_t *__restrict__ q,
                     SsspCounters *cnt, unsigned cap) {
  const unsigned w = blockIdx.x * GDN_BLOCK + threadIdx.x;
  unsigned word = (w < nwords) ? bits[w] : 0u;
  const unsigned n = __popc(word);
  const unsigned incl = gdn_wave_incl_scan(n);
  const unsigned total = __shfl(incl, 63, 64);
  if (total == 0) return;
  unsigned base = 0;
  if (gdn_lane() == 63) base = atomicAdd(&cnt->near_count, total);
  base = __shfl(base, 63, 64);
  unsigned pos = base + incl - n;
  while (word) {
    const int k = __ffs((int)word) - 1;
    word &= word - 1u;
    const unsigned v = w * 32u + (unsigned)k;
    if (v < (unsigned)m) {
      if (pos < cap) q[pos] = (vid_t)v;
      else cnt->overflow = 1u;
      pos++;
    }
  }
}

struct gdn_sssp_plan {
  const gdn_graph *g = nullptr;
  const int32_t *d_weight = nullptr;
  bool dense = false;
  PbPlan pb;               // of the OUT-CSR (rows are sources), no fp32 vals
  DevBuf<float> Wp;        // weights in tile order (int32 bits)
  DevBuf<unsigned> cand;   // candidate distances, bin-major
  DevBuf<unsigned> improved;
  DevBuf<vid_t> near0, near1, far0, far1;
  DevBuf<int32_t> stamp;
  DevBuf<unsigned> in_far;
  DevBuf<unsigned long long> bigitems;
  DevBuf<SsspCounters> cnt;
  unsigned cap = 0, bigcap = 0, nwords = 0;
  double prep_ms = 0;
};

static int sssp_plan_init(gdn_sssp_plan &p, const gdn_graph *g, const int32_t *d_weight, bool dense) {
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
  if (dense && g->nnz > 0) {
    int lg = 10;
    while (lg < 15 && ((int64_t)1 << (lg + 9)) < (int64_t)m) lg++;
    GDN_TRY(pb_build(g, m, lg, lg, p.pb, /*alloc_vals=*/false, reinterpret_cast<const float *>(d_weight), &p.Wp,
                     /*compact=*/false, /*rows_are_sources=*/true,
                     /*pad=*/getenv("GDN_SSSP_PAD") ? (unsigned)atoi(getenv("GDN_SSSP_PAD")) : 32u, /*log_group=*/3));
    GDN_TRY(p.cand.alloc(p.pb.n_pad + 8));
    // slots in the alignment gaps of the layout are never written by phase A: keep them neutral
    GDN_TRY(gdn_fill_i32(reinterpret_cast<int32_t *>(p.cand.p), GDN_DIST_INF, (size_t)p.pb.n_pad + 8, 0));
    p.nwords = (unsigned)(((uint64_t)p.pb.nbins << lg) / 32u);
    GDN_TRY(p.improved.alloc(p.nwords + 64));
    const int lds = (int)((sizeof(unsigned) << lg) + 16);
    hipError_t e = hipFuncSetAttribute((const void *)sssp_pb_expand_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)sssp_pb_accumulate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      gdn_set_error("hipFuncSetAttribute(dynamic LDS): %s", hipGetErrorString(e));
      return GDN_ERR_HIP;
    }
    p.dense = true;
  }
  GDN_HIP(hipDeviceSynchronize());
  p.prep_ms = t.stop_ms();
  return GDN_OK;
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
  SsspCounters h;
  ExpBigList big;
  big.items = p.bigitems.p;
  big.capacity = p.bigcap;
  const unsigned cap = p.cap;
  auto clamp = [](int64_t x) { return (int32_t)(x > GDN_DIST_INF ? GDN_DIST_INF : x); };
  // dense sweeps start when the NEAR list owns more than nnz / dense_in out-edges and go on while more than m / dense_out
  // rows improve per sweep (tuning knobs)
  unsigned long long dense_in = 24, dense_out = 16;  // measured on RMAT-24, U[1,255]: m/256 -> m/16 takes 7.2 / 9.2 ms to 6.2 / 7.5 ms
  if (const char *e = getenv("GDN_SSSP_DENSE_IN")) dense_in = atoi(e) > 0 ? (unsigned long long)atoi(e) : dense_in;
  if (const char *e = getenv("GDN_SSSP_DENSE_OUT")) dense_out = atoi(e) > 0 ? (unsigned long long)atoi(e) : dense_out;
  for (;;) {
    while (n_near > 0) {
      if (p.dense && near_edges * dense_in > (unsigned long long)g->nnz) {
        // ---- heavy frontier: Bellman-Ford sweeps over all edges until few rows still improve
        const size_t lds = (sizeof(unsigned) << p.pb.log_chunk) + 16;
        unsigned long long improved = 0;
        do {
          ++phases;
          memset(&h, 0, sizeof(h));
          h.min_far = GDN_DIST_INF;
          GDN_HIP(hipMemcpyAsync(p.cnt.p, &h, sizeof(h), hipMemcpyHostToDevice, 0));
          hipLaunchKernelGGL(sssp_pb_expand_kernel, dim3(p.pb.nchunks), dim3(PB_THREADS), lds, 0, d_dist, m,
                             p.pb.log_chunk, p.pb.chunk_ptr.p, p.pb.chunk_order.p, p.pb.U.p, p.pb.G.p,
                             reinterpret_cast<const uint32_t *>(p.Wp.p), p.cand.p);
          hipLaunchKernelGGL(sssp_pb_accumulate_kernel, dim3(p.pb.nbins), dim3(PB_THREADS), lds, 0, m, p.pb.log_bin,
                             p.pb.bin_ptr.p, p.pb.bin_order.p, p.pb.V.p, p.cand.p, d_dist, p.improved.p, p.cnt.p);
          GDN_HIP(hipMemcpy(&h, p.cnt.p, sizeof(h), hipMemcpyDeviceToHost));
          improved = h.relaxed;
        } while (improved * dense_out > (unsigned long long)m);
        // the rows improved by the LAST sweep are the only ones with unpropagated distances: they
        // become a plain Bellman-Ford worklist (one infinite bucket); the parked FAR list is
        // covered by the sweeps and dropped
        memset(&h, 0, sizeof(h));
        h.min_far = GDN_DIST_INF;
        GDN_HIP(hipMemcpyAsync(p.cnt.p, &h, sizeof(h), hipMemcpyHostToDevice, 0));
        GDN_HIP(hipMemsetAsync(p.in_far.p, 0, (size_t)m * 4, 0));
        hipLaunchKernelGGL(sssp_bitmap_to_queue, dim3(gdn_nblocks(p.nwords)), dim3(GDN_BLOCK), 0, 0, p.improved.p,
                           p.nwords, m, near_in, p.cnt.p, cap);
        GDN_HIP(hipMemcpy(&h, p.cnt.p, sizeof(h), hipMemcpyDeviceToHost));
        n_near = h.near_count;
        n_far = 0;
        near_edges = 0;  // at most m / dense_out rows: back to the worklist
        thr_lo = 0;
        thr_hi = (int64_t)GDN_DIST_INF;
        continue;
      }
      ++pass;
      ++phases;
      h.near_count = 0;
      h.far_count = n_far;
      h.big_count = 0;
      h.overflow = 0;
      h.min_far = GDN_DIST_INF;
      h.pad = 0;
      h.relaxed = 0;
      GDN_HIP(hipMemcpyAsync(p.cnt.p, &h, sizeof(h), hipMemcpyHostToDevice, 0));
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
      vis.du = 0;
      big.count = &p.cnt.p->big_count;
      big.overflow = &p.cnt.p->overflow;
      hipLaunchKernelGGL(sssp_relax_kernel, dim3(gdn_nblocks(n_near)), dim3(GDN_BLOCK), 0, 0, g->rowptr, near_in,
                         n_near, clamp(thr_lo), big, vis);
      hipLaunchKernelGGL(sssp_relax_big_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
      GDN_HIP(hipMemcpy(&h, p.cnt.p, sizeof(h), hipMemcpyDeviceToHost));
      if (h.overflow) {
        gdn_set_error("gdn_sssp: device worklist overflow");
        return GDN_ERR_OVERFLOW;
      }
      n_near = h.near_count;
      n_far = h.far_count;
      near_edges = h.relaxed;
      vid_t *t = near_in;
      near_in = near_out;
      near_out = t;
    }
    if (n_far == 0) break;
    // ---- next non-empty bucket (omp_base.cc:66-72 votes for the smallest non-empty bin)
    h.near_count = 0;
    h.far_count = 0;
    h.big_count = 0;
    h.overflow = 0;
    h.min_far = GDN_DIST_INF;
    h.relaxed = 0;
    GDN_HIP(hipMemcpyAsync(p.cnt.p, &h, sizeof(h), hipMemcpyHostToDevice, 0));
    hipLaunchKernelGGL(sssp_far_min_kernel, dim3(gdn_nblocks(n_far) < 2048u ? gdn_nblocks(n_far) : 2048u), dim3(GDN_BLOCK), 0, 0,
                       far_cur, n_far, d_dist,
                       clamp(thr_hi), p.cnt.p);
    GDN_HIP(hipMemcpy(&h, p.cnt.p, sizeof(h), hipMemcpyDeviceToHost));
    if (h.min_far == GDN_DIST_INF) break;  // only stale entries were left
    const int64_t old_hi = thr_hi;
    thr_lo = ((int64_t)h.min_far / delta) * (int64_t)delta;
    thr_hi = thr_lo + delta;
    hipLaunchKernelGGL(sssp_far_split_kernel, dim3(gdn_nblocks(n_far) < 2048u ? gdn_nblocks(n_far) : 2048u), dim3(GDN_BLOCK), 0, 0,
                       far_cur, n_far, d_dist,
                       clamp(old_hi), clamp(thr_hi), p.in_far.p, near_in, far_nxt, p.cnt.p, cap, g->rowptr);
    GDN_HIP(hipMemcpy(&h, p.cnt.p, sizeof(h), hipMemcpyDeviceToHost));
    if (h.overflow) {
      gdn_set_error("gdn_sssp: device worklist overflow");
      return GDN_ERR_OVERFLOW;
    }
    n_near = h.near_count;
    n_far = h.far_count;
    near_edges = h.relaxed;
    vid_t *t = far_cur;
    far_cur = far_nxt;
    far_nxt = t;
  }
  GDN_HIP(hipGetLastError());
  st.solve_ms = tsolve.stop_ms();
  st.iterations = phases;
  uint64_t te = 0;
  GDN_TRY(gdn_reached_edges(g, d_dist, GDN_DIST_INF, &te));
  st.edges_traversed = te;
  if (stats) *stats = st;
  return GDN_OK;
}

extern "C" {

int gdn_sssp_plan_create(const gdn_graph *g, const int32_t *d_weight, int32_t dense, gdn_sssp_plan **plan) {
  GDN_REQUIRE(plan != nullptr, "plan");
  *plan = nullptr;
  GDN_REQUIRE(g != nullptr && (d_weight != nullptr || g->nnz == 0), "graph / d_weight");
  gdn_sssp_plan *p = new gdn_sssp_plan();
  const int rc = sssp_plan_init(*p, g, d_weight, dense != 0);
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
  return sssp_run(*plan, source, delta, d_dist, stats);
}

int gdn_sssp_dev(const gdn_graph *g, const int32_t *d_weight, int32_t source, int32_t delta, int32_t *d_dist,
                 gdn_stats *stats) {
  GDN_REQUIRE(g != nullptr && d_dist != nullptr && (d_weight != nullptr || g->nnz == 0), "null argument");
  GDN_REQUIRE(source >= 0 && source < g->m, "source out of range");
  GDN_REQUIRE(delta >= 1, "delta must be >= 1");
  gdn_sssp_plan p;
  GDN_TRY(sssp_plan_init(p, g, d_weight, /*dense=*/false));
  return sssp_run(p, source, delta, d_dist, stats);
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
