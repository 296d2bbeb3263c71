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
#include <string.h>

#include "gdn_expand.hpp"

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
          if (nd < thr_hi) push_near = atomicExch(&stamp[dst], pass) != pass;
          else push_far = atomicExch(&in_far[dst], 1u) == 0u;
        }
      }
    }
    gdn_wl_push(near_out, &cnt->near_count, cap, push_near, dst, &cnt->overflow);
    gdn_wl_push(far_out, &cnt->far_count, cap, push_far, dst, &cnt->overflow);
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
sssp_relax_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ near_in, unsigned n,
                  int32_t thr_lo, ExpBigList big, SsspVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
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
  gdn_expand_wave(b, e, v, big, vis, s_scan[threadIdx.x >> 6]);
}

__global__ void __launch_bounds__(GDN_BLOCK)
sssp_relax_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, SsspVis vis) {
  vis.du = 0;
  gdn_expand_big_items(rowptr, big, vis);
}

// smallest distance parked in FAR that is still >= thr_hi (stale entries are ignored)
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_far_min_kernel(const vid_t *__restrict__ far_in, unsigned n, const int32_t *__restrict__ dist,
                    int32_t thr_hi, SsspCounters *cnt) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  int32_t d = GDN_DIST_INF;
  if (i < n) {
    const int32_t x = dist[far_in[i]];
    if (x >= thr_hi) d = x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int32_t t = __shfl_xor(d, o, 64);
    d = t < d ? t : d;
  }
  if (gdn_lane() == 0 && d != GDN_DIST_INF) atomicMin(&cnt->min_far, d);
}

// FAR -> {NEAR of the new bucket, FAR kept, dropped}
__global__ void __launch_bounds__(GDN_BLOCK)
sssp_far_split_kernel(const vid_t *__restrict__ far_in, unsigned n, const int32_t *__restrict__ dist,
                      int32_t old_hi, int32_t new_hi, unsigned *__restrict__ in_far, vid_t *__restrict__ near_out,
                      vid_t *__restrict__ far_out, SsspCounters *cnt, unsigned cap) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  bool to_near = false, to_far = false;
  vid_t w = 0;
  if (i < n) {
    w = far_in[i];
    const int32_t d = dist[w];
    if (d >= new_hi) to_far = true;
    else {
      in_far[w] = 0u;
      to_near = d >= old_hi;
    }
  }
  gdn_wl_push(near_out, &cnt->near_count, cap, to_near, w, &cnt->overflow);
  gdn_wl_push(far_out, &cnt->far_count, cap, to_far, w, &cnt->overflow);
}

__global__ void sssp_seed_kernel(int32_t source, int32_t *dist, vid_t *near) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    dist[source] = 0;
    near[0] = source;
  }
}

int gdn_reached_edges(const gdn_graph *g, const int32_t *d_dist, int32_t unreached, uint64_t *out);

extern "C" {

int gdn_sssp_dev(const gdn_graph *g, const int32_t *d_weight, int32_t source, int32_t delta, int32_t *d_dist,
                 gdn_stats *stats) {
  GDN_REQUIRE(g != nullptr && d_dist != nullptr && (d_weight != nullptr || g->nnz == 0), "null argument");
  GDN_REQUIRE(source >= 0 && source < g->m, "source out of range");
  GDN_REQUIRE(delta >= 1, "delta must be >= 1");
  const int32_t m = g->m;
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  HostTimer tprep, tsolve;
  tprep.start();
  DevBuf<vid_t> near0, near1, far0, far1;
  DevBuf<int32_t> stamp;
  DevBuf<unsigned> in_far;
  DevBuf<unsigned long long> bigitems;
  DevBuf<SsspCounters> cnt;
  const unsigned cap = (unsigned)m;
  const uint64_t bigcap64 = g->nnz / EXP_CHUNK + (uint64_t)m / 64 + 1024;
  const unsigned bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
  GDN_TRY(near0.alloc(cap));
  GDN_TRY(near1.alloc(cap));
  GDN_TRY(far0.alloc(cap));
  GDN_TRY(far1.alloc(cap));
  GDN_TRY(stamp.alloc(m));
  GDN_TRY(in_far.alloc(m));
  GDN_TRY(bigitems.alloc(bigcap));
  GDN_TRY(cnt.alloc(1));
  st.prep_ms = tprep.stop_ms();

  tsolve.start();  // omp_base.cc:27 t.Start()
  GDN_TRY(gdn_fill_i32(d_dist, GDN_DIST_INF, (size_t)m, 0));
  GDN_HIP(hipMemsetAsync(stamp.p, 0, (size_t)m * 4, 0));
  GDN_HIP(hipMemsetAsync(in_far.p, 0, (size_t)m * 4, 0));
  hipLaunchKernelGGL(sssp_seed_kernel, dim3(1), dim3(64), 0, 0, source, d_dist, near0.p);
  vid_t *near_in = near0.p, *near_out = near1.p, *far_cur = far0.p, *far_nxt = far1.p;
  unsigned n_near = 1, n_far = 0;
  int64_t thr_lo = 0, thr_hi = delta;
  int32_t pass = 0;
  int phases = 0;
  SsspCounters h;
  ExpBigList big;
  big.items = bigitems.p;
  big.capacity = bigcap;
  auto clamp = [](int64_t x) { return (int32_t)(x > GDN_DIST_INF ? GDN_DIST_INF : x); };
  for (;;) {
    while (n_near > 0) {
      ++pass;
      ++phases;
      h.near_count = 0;
      h.far_count = n_far;
      h.big_count = 0;
      h.overflow = 0;
      h.min_far = GDN_DIST_INF;
      h.pad = 0;
      h.relaxed = 0;
      GDN_HIP(hipMemcpyAsync(cnt.p, &h, sizeof(h), hipMemcpyHostToDevice, 0));
      SsspVis vis;
      vis.colidx = g->colidx;
      vis.weight = d_weight;
      vis.dist = d_dist;
      vis.stamp = stamp.p;
      vis.in_far = in_far.p;
      vis.near_out = near_out;
      vis.far_out = far_cur;  // FAR grows in place behind its current tail
      vis.cnt = cnt.p;
      vis.cap = cap;
      vis.thr_hi = clamp(thr_hi);
      vis.pass = pass;
      vis.du = 0;
      big.count = &cnt.p->big_count;
      big.overflow = &cnt.p->overflow;
      hipLaunchKernelGGL(sssp_relax_kernel, dim3(gdn_nblocks(n_near)), dim3(GDN_BLOCK), 0, 0, g->rowptr, near_in,
                         n_near, clamp(thr_lo), big, vis);
      hipLaunchKernelGGL(sssp_relax_big_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
      GDN_HIP(hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost));
      if (h.overflow) {
        gdn_set_error("gdn_sssp: device worklist overflow");
        return GDN_ERR_OVERFLOW;
      }
      n_near = h.near_count;
      n_far = h.far_count;
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
    GDN_HIP(hipMemcpyAsync(cnt.p, &h, sizeof(h), hipMemcpyHostToDevice, 0));
    hipLaunchKernelGGL(sssp_far_min_kernel, dim3(gdn_nblocks(n_far)), dim3(GDN_BLOCK), 0, 0, far_cur, n_far, d_dist,
                       clamp(thr_hi), cnt.p);
    GDN_HIP(hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost));
    if (h.min_far == GDN_DIST_INF) break;  // only stale entries were left
    const int64_t old_hi = thr_hi;
    thr_lo = ((int64_t)h.min_far / delta) * (int64_t)delta;
    thr_hi = thr_lo + delta;
    hipLaunchKernelGGL(sssp_far_split_kernel, dim3(gdn_nblocks(n_far)), dim3(GDN_BLOCK), 0, 0, far_cur, n_far, d_dist,
                       clamp(old_hi), clamp(thr_hi), in_far.p, near_in, far_nxt, cnt.p, cap);
    GDN_HIP(hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost));
    if (h.overflow) {
      gdn_set_error("gdn_sssp: device worklist overflow");
      return GDN_ERR_OVERFLOW;
    }
    n_near = h.near_count;
    n_far = h.far_count;
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
