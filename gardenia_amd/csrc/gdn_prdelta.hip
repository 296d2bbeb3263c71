// gdn_prdelta.hip -- delta PageRank (src/pr/delta.cu:140-202, OpenMP twin src/pr/omp_delta.cc:52-107).
//
// What the reference computes: scores start at 1/m, deltas at 1/m.  While the frontier (the vertices whose last delta
// was larger than epsilon2 * score) holds at least m / push_div vertices an iteration PULLS contrib = delta / out-degree
// of ALL vertices over the in-CSR (delta.cu:182-184), otherwise it PUSHES the frontier's deltas over the out-CSR with
// atomic adds (delta.cu:179-180); then delta = d * sum (first iteration: base + d * sum - 1/m), score += delta,
// sum = 0, and the L1 norm of the deltas is the convergence test (delta.cu:187-197).
//
// How it runs here:
//   * pull = one SpMV with the pattern matrix of the in-CSR on the SpMV plan of gdn_spmv.hip: the value-free form of the
//     propagation-blocked layout with its record tiers for graphs above 2^22 edges, whose SIGNED fixed-point
//     accumulation takes the negative deltas the unsigned PageRank layout of gdn_pr.hip cannot; the result is the
//     exactly-summed, once-rounded row sum (order independent, run-to-run identical);
//   * push = what the reference's push computes -- sums[dst] = sum of the contributions of dst's in-neighbours IN THE
//     FRONTIER -- in one of two ways, by the number of out-edges the frontier has (counted by the update kernel):
//     a heavy frontier (>= nnz / 64 edges; GDN_PRD_PUSH_DIV) runs as the same pull with the contributions of the
//     vertices outside the frontier masked to 0 -- 1.3 ms at RMAT-25 whatever the frontier, deterministic, where 3.4 M
//     frontier vertices took 16 ms of fp32 atomics --; a light one runs the wave64 neighbour expansion of
//     gdn_expand.hpp over the vertices flagged active with hardware fp32 atomic adds (global_atomic_add_f32) like the
//     reference's atomicAdd, the one order-dependent step, as in the reference;
//   * update + frontier + L1 norm are ONE kernel (the reference: update, Worklist2 push per vertex, l1norm): the frontier
//     is a byte flag per vertex, not a queue -- a queue costs one hot-counter atomic per wave of vertices (~12 ns each on
//     this chip, 25 ms at m = 2^27) and the push kernel's scan of m flags costs 0.03 ms --, counts and the norm go through
//     per-workgroup partials summed in a fixed order, so the trace is deterministic.
#include "gardenia_hip.h"
#include "gdn_common.hpp"
#include "gdn_expand.hpp"

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#define PRD_GRID 2048  // workgroups of the update kernel == number of partials

struct PrdCounters {
  double diff;
  unsigned long long items;
  unsigned long long edges;  // out-edges of the frontier
  unsigned big_count;
  unsigned overflow;
};

struct gdn_pr_delta_plan {
  const gdn_graph *gin = nullptr, *gout = nullptr;
  gdn_spmv_plan *sp = nullptr;
  int32_t layout = 0;
  DevBuf<float> ones;  // CSR layout: the pattern's values (the PB layout is built value-free)
  DevBuf<int32_t> deg;
  DevBuf<float> sums, contrib, masked;  // masked: allocated by the first push that runs as a pull
  DevBuf<uint8_t> active;
  DevBuf<double> pdiff;
  DevBuf<unsigned> pitems;
  DevBuf<unsigned long long> pedges;
  DevBuf<PrdCounters> cnt;
  DevBuf<unsigned long long> bigitems;
  unsigned bigcap = 0;
  std::vector<double> tr_diff;
  std::vector<int32_t> tr_items, tr_mode;
  ~gdn_pr_delta_plan() {
    if (sp) gdn_spmv_plan_free(sp);
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
prd_degree_kernel(const eoff_t *__restrict__ rowptr, int32_t m, int32_t *__restrict__ deg) {
  const size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (size_t)m) deg[v] = (int32_t)(rowptr[v + 1] - rowptr[v]);
}

__global__ void __launch_bounds__(GDN_BLOCK)
prd_fill_kernel(float *__restrict__ p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * GDN_BLOCK) p[i] = v;
}

// delta.cu:15-22 with the first contrib (delta = 1/m).  A vertex without out-edges is no row's source: its quotient (inf
// or nan in the reference) is never read, and 0 keeps it out of the max |x| the fixed-point scale of the pull is taken from
__global__ void __launch_bounds__(GDN_BLOCK)
prd_init_kernel(int32_t m, float *__restrict__ sums, float *__restrict__ contrib, const int32_t *__restrict__ deg, float init_score) {
  const size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (size_t)m) {
    sums[v] = 0.0f;
    const int32_t d = deg[v];
    contrib[v] = d ? __fdiv_rn(init_score, (float)d) : 0.0f;
  }
}

// delta.cu:40-45 (contrib = delta / degree) is part of the update kernel below.  A push that runs as a pull of the
// frontier's terms reads this masked copy instead: only the frontier contributes
__global__ void __launch_bounds__(GDN_BLOCK)
prd_mask_kernel(const float *__restrict__ contrib, const uint8_t *__restrict__ active, int32_t m, float *__restrict__ masked) {
  const size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (size_t)m) masked[v] = active[v] ? contrib[v] : 0.0f;
}

// delta.cu:24-38 on the expansion tiers of gdn_expand.hpp
struct PrdPushVis {
  const vid_t *__restrict__ colidx;
  const float *__restrict__ contrib;  // delta / out-degree, written by the update kernel
  float *__restrict__ sums;
  float c;
  int big;
  __device__ __forceinline__ void begin_big(vid_t v) {
    big = 1;
    c = contrib[v];
  }
  __device__ __forceinline__ void edge(int owner, eoff_t k, bool valid) {
    const float cc = big ? c : __shfl(c, owner, 64);
    if (valid) unsafeAtomicAdd(&sums[__builtin_nontemporal_load(colidx + k)], cc);
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
prd_push_kernel(const eoff_t *__restrict__ rowptr, int32_t m, const uint8_t *__restrict__ active, ExpBigList big, PrdPushVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vis.c = 0.0f;
  vis.big = 0;
  if (v < (unsigned)m && active[v]) {
    b = rowptr[v];
    e = rowptr[v + 1];
    vis.c = vis.contrib[v];
  }
  gdn_expand_wave(b, e, (vid_t)v, big, vis, s_scan[threadIdx.x >> 6]);
}

__global__ void __launch_bounds__(GDN_BLOCK)
prd_push_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, PrdPushVis vis) {
  vis.big = 1;
  vis.c = 0.0f;
  gdn_expand_big_items(rowptr, big, vis);
}

// delta.cu:103-129 (update_first / update), the frontier test and l1norm (:131-138) in one pass
template <bool FIRST>
__global__ void __launch_bounds__(GDN_BLOCK)
prd_update_kernel(int32_t m, float *__restrict__ scores, float *__restrict__ sums, float *__restrict__ contrib,
                  uint8_t *__restrict__ active, const int32_t *__restrict__ deg, float base_score, float init_score,
                  float damping, float epsilon2, double *__restrict__ pdiff, unsigned *__restrict__ pitems,
                  unsigned long long *__restrict__ pedges) {
  __shared__ double s_d[GDN_WAVES_PER_BLOCK];
  __shared__ unsigned s_n[GDN_WAVES_PER_BLOCK];
  __shared__ unsigned long long s_e[GDN_WAVES_PER_BLOCK];
  unsigned long long edges = 0;
  // a workgroup owns ONE contiguous range (fixed by m and the grid): its partial does not depend on scheduling
  const size_t per = (((size_t)m + PRD_GRID - 1) / PRD_GRID + 3) & ~(size_t)3;
  const size_t lo = (size_t)blockIdx.x * per;
  const size_t hi = lo + per < (size_t)m ? lo + per : (size_t)m;
  double diff = 0.0;
  unsigned items = 0;
  for (size_t u = lo + threadIdx.x; u < hi; u += GDN_BLOCK) {
    float d = gdn_fmul(damping, sums[u]);
    if (FIRST) d = gdn_fsub(gdn_fadd(base_score, d), init_score);
    const float s = gdn_fadd(scores[u], d);
    const int32_t dg = deg[u];
    contrib[u] = dg ? __fdiv_rn(d, (float)dg) : 0.0f;  // the next iteration's contribution (delta.cu:43, :32)
    scores[u] = s;
    sums[u] = 0.0f;
    const bool a = fabsf(d) > gdn_fmul(epsilon2, s);
    active[u] = a ? 1 : 0;
    items += a ? 1u : 0u;
    edges += a ? (unsigned long long)(unsigned)dg : 0ull;
    diff += (double)fabsf(d);
  }
  diff = gdn_block_sum(diff, s_d);
  items = gdn_block_sum(items, s_n);
  edges = gdn_block_sum(edges, s_e);
  if (threadIdx.x == 0) {
    pdiff[blockIdx.x] = diff;
    pitems[blockIdx.x] = items;
    pedges[blockIdx.x] = edges;
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
prd_reduce_kernel(const double *__restrict__ pdiff, const unsigned *__restrict__ pitems,
                  const unsigned long long *__restrict__ pedges, PrdCounters *__restrict__ out) {
  __shared__ double s_d[GDN_WAVES_PER_BLOCK];
  __shared__ unsigned long long s_n[GDN_WAVES_PER_BLOCK];
  double d = 0.0;
  unsigned long long n = 0, e = 0;
  for (unsigned i = threadIdx.x; i < PRD_GRID; i += GDN_BLOCK) {
    d += pdiff[i];
    n += pitems[i];
    e += pedges[i];
  }
  d = gdn_block_sum(d, s_d);
  n = gdn_block_sum(n, s_n);
  e = gdn_block_sum(e, s_n);
  if (threadIdx.x == 0) {
    out->diff = d;
    out->items = n;
    out->edges = e;
  }
}

extern "C" {

int gdn_pr_delta_plan_create(const gdn_graph *in_csr, const gdn_graph *out_csr, int32_t layout, gdn_pr_delta_plan **plan) {
  GDN_REQUIRE(in_csr != nullptr && out_csr != nullptr && plan != nullptr, "graphs / plan");
  GDN_REQUIRE(in_csr->m == out_csr->m && in_csr->nnz == out_csr->nnz, "in-CSR and out-CSR of different graphs");
  GDN_REQUIRE(layout == GDN_LAYOUT_AUTO || layout == GDN_LAYOUT_CSR || layout == GDN_LAYOUT_PB, "layout");
  const int32_t m = in_csr->m;
  const uint64_t nnz = in_csr->nnz;
  gdn_pr_delta_plan *p = new (std::nothrow) gdn_pr_delta_plan();
  if (!p) return GDN_ERR_OOM;
  p->gin = in_csr;
  p->gout = out_csr;
  int rc = GDN_OK;
  do {
    if (layout == GDN_LAYOUT_AUTO) {
      const char *env = gdn_option("GDN_PRD_LAYOUT");  // test knob: 'p' / 'c' force the layout of the pull's plan
      if (env && env[0] == 'p') layout = GDN_LAYOUT_PB;
      else if (env && env[0] == 'c') layout = GDN_LAYOUT_CSR;
      else layout = nnz >= (1ull << 22) ? GDN_LAYOUT_PB : GDN_LAYOUT_CSR;
    }
    p->layout = layout;
    if (layout == GDN_LAYOUT_CSR) {  // the merge-path kernel reads a value per nonzero; the PB layout has a pattern form
      if ((rc = p->ones.alloc(nnz ? nnz : 1))) break;
      hipLaunchKernelGGL(prd_fill_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, p->ones.p, (size_t)nnz, 1.0f);
    }
    if ((rc = gdn_spmv_plan_create(in_csr, p->ones.p, layout, &p->sp))) break;
    if ((rc = p->deg.alloc(m)) || (rc = p->sums.alloc(m)) || (rc = p->contrib.alloc(m)) ||
        (rc = p->active.alloc(m)) || (rc = p->pdiff.alloc(PRD_GRID)) || (rc = p->pitems.alloc(PRD_GRID)) || (rc = p->pedges.alloc(PRD_GRID)) ||
        (rc = p->cnt.alloc(1)))
      break;
    const uint64_t bigcap64 = nnz / EXP_CHUNK + (uint64_t)m / 64 + 1024;
    p->bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
    if ((rc = p->bigitems.alloc(p->bigcap))) break;
    hipLaunchKernelGGL(prd_degree_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, out_csr->rowptr, m, p->deg.p);
    if (hipMemset(p->cnt.p, 0, sizeof(PrdCounters)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      gdn_set_error("gdn_pr_delta_plan_create: %s", hipGetErrorString(hipGetLastError()));
      rc = GDN_ERR_HIP;
      break;
    }
  } while (0);
  if (rc) {
    delete p;
    return rc;
  }
  *plan = p;
  return GDN_OK;
}

int gdn_pr_delta_plan_free(gdn_pr_delta_plan *plan) {
  delete plan;
  return GDN_OK;
}

int gdn_pr_delta_run(gdn_pr_delta_plan *plan, float *d_scores, float damping, double epsilon, float epsilon2,
                     int32_t max_iter, int32_t push_div, gdn_stats *stats) {
  GDN_REQUIRE(plan != nullptr && d_scores != nullptr, "plan / d_scores");
  GDN_REQUIRE(max_iter >= 1 && push_div >= 1, "max_iter / push_div");
  gdn_pr_delta_plan &p = *plan;
  const int32_t m = p.gin->m;
  const float base_score = (1.0f - damping) / m;  // delta.cu:166
  const float init_score = 1.0f / m;              // delta.cu:167
  const unsigned nb = gdn_nblocks((uint64_t)m);
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  p.tr_diff.clear();
  p.tr_items.clear();
  p.tr_mode.clear();
  HostTimer tsolve;
  hipLaunchKernelGGL(prd_init_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, m, p.sums.p, p.contrib.p, p.deg.p, init_score);
  GDN_HIP(hipDeviceSynchronize());
  tsolve.start();  // delta.cu:174
  ExpBigList big;
  big.items = p.bigitems.p;
  big.capacity = p.bigcap;
  big.count = &p.cnt.p->big_count;
  big.overflow = &p.cnt.p->overflow;
  long long nitems = m;
  unsigned long long fedges = p.gin->nnz;  // out-edges of the frontier
  unsigned long long heavy_div = 64;
  if (const char *e = gdn_option("GDN_PRD_PUSH_DIV")) heavy_div = strtoull(e, nullptr, 10) ? strtoull(e, nullptr, 10) : 1;
  int iter = 0;
  uint64_t pull_iters = 0;
  PrdCounters h;
  memset(&h, 0, sizeof(h));
  do {
    ++iter;
    const bool push = nitems < (long long)(m / push_div);  // delta.cu:178 (8), omp_delta.cc:69 (10)
    const bool masked = push && fedges * heavy_div >= p.gin->nnz;  // heavy frontier: the pull, frontier terms only
    if (push && !masked) {
      PrdPushVis vis;
      vis.colidx = p.gout->colidx;
      vis.contrib = p.contrib.p;
      vis.sums = p.sums.p;
      vis.c = 0.0f;
      vis.big = 0;
      hipLaunchKernelGGL(prd_push_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, p.gout->rowptr, m, p.active.p, big, vis);
      hipLaunchKernelGGL(prd_push_big_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, p.gout->rowptr, big, vis);
      GDN_HIP(hipMemsetAsync(&p.cnt.p->big_count, 0, sizeof(unsigned), 0));
    } else {
      const float *x = p.contrib.p;
      if (masked) {
        if (!p.masked.p) GDN_TRY(p.masked.alloc(m));
        hipLaunchKernelGGL(prd_mask_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, p.contrib.p, p.active.p, m, p.masked.p);
        x = p.masked.p;
      }
      GDN_TRY(gdn_spmv_dev(p.sp, p.ones.p, x, p.sums.p, nullptr));  // sums are 0 here: sums += A x
      pull_iters++;
    }
    if (iter == 1)
      hipLaunchKernelGGL(prd_update_kernel<true>, dim3(PRD_GRID), dim3(GDN_BLOCK), 0, 0, m, d_scores, p.sums.p, p.contrib.p,
                         p.active.p, p.deg.p, base_score, init_score, damping, epsilon2, p.pdiff.p, p.pitems.p, p.pedges.p);
    else
      hipLaunchKernelGGL(prd_update_kernel<false>, dim3(PRD_GRID), dim3(GDN_BLOCK), 0, 0, m, d_scores, p.sums.p, p.contrib.p,
                         p.active.p, p.deg.p, base_score, init_score, damping, epsilon2, p.pdiff.p, p.pitems.p, p.pedges.p);
    hipLaunchKernelGGL(prd_reduce_kernel, dim3(1), dim3(GDN_BLOCK), 0, 0, p.pdiff.p, p.pitems.p, p.pedges.p, p.cnt.p);
    GDN_HIP(hipGetLastError());
    GDN_HIP(hipMemcpy(&h, p.cnt.p, sizeof(h), hipMemcpyDeviceToHost));  // delta.cu:193,195: one read back per iteration
    if (h.overflow) {
      gdn_set_error("gdn_pr_delta_run: device worklist overflow");
      return GDN_ERR_OVERFLOW;
    }
    nitems = (long long)h.items;
    fedges = h.edges;
    p.tr_diff.push_back(h.diff);
    p.tr_items.push_back((int32_t)nitems);
    p.tr_mode.push_back(push ? (masked ? 3 : 1) : 0);
    if (h.diff < epsilon) break;  // delta.cu:197
  } while (nitems > 0 && iter < max_iter);
  GDN_HIP(hipDeviceSynchronize());
  st.solve_ms = tsolve.stop_ms();
  GDN_TRY(gdn_spmv_plan_check(p.sp));
  st.iterations = iter;  // delta.cu:200 (omp_delta.cc:105 prints iter + 1)
  st.last_error = h.diff;
  st.edges_traversed = p.gin->nnz * pull_iters;  // sweeps over all edges; the atomic pushes' edges are not counted
  if (stats) *stats = st;
  return GDN_OK;
}

int gdn_pr_delta_trace(const gdn_pr_delta_plan *plan, int32_t capacity, int32_t *n, double *diff, int32_t *items, int32_t *mode) {
  GDN_REQUIRE(plan != nullptr && n != nullptr, "plan / n");
  const int32_t have = (int32_t)plan->tr_diff.size();
  *n = have;
  for (int32_t i = 0; i < have && i < capacity; i++) {
    if (diff) diff[i] = plan->tr_diff[i];
    if (items) items[i] = plan->tr_items[i];
    if (mode) mode[i] = plan->tr_mode[i];
  }
  return GDN_OK;
}

// Host API: one call == the PRSolver of src/pr/delta.cu:140 / src/pr/omp_delta.cc:52 (out-degrees = out-CSR row lengths,
// which is what both read them as).
int gdn_pr_delta(int32_t m, uint64_t nnz, const uint64_t *in_rowptr, const int32_t *in_colidx, const uint64_t *out_rowptr,
                 const int32_t *out_colidx, float *scores, float damping, double epsilon, float epsilon2, int32_t max_iter,
                 int32_t push_div, gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && in_rowptr && out_rowptr && scores && ((in_colidx && out_colidx) || nnz == 0), "null argument");
  GDN_TRY(gdn_require_device());
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  HostTimer th2d, tprep;
  th2d.start();
  gdn_graph *gi = nullptr, *go = nullptr;
  gdn_pr_delta_plan *plan = nullptr;
  DevBuf<float> d_scores;
  int rc = GDN_OK;
  do {
    if ((rc = gdn_graph_upload(m, nnz, in_rowptr, in_colidx, &gi))) break;
    if ((rc = gdn_graph_upload(m, nnz, out_rowptr, out_colidx, &go))) break;
    if ((rc = d_scores.alloc(m))) break;
    if (hipMemcpy(d_scores.p, scores, (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess) {
      gdn_set_error("gdn_pr_delta: upload failed");
      rc = GDN_ERR_HIP;
      break;
    }
    const double h2d = th2d.stop_ms();
    tprep.start();
    if ((rc = gdn_pr_delta_plan_create(gi, go, GDN_LAYOUT_AUTO, &plan))) break;
    const double prep = tprep.stop_ms();
    if ((rc = gdn_pr_delta_run(plan, d_scores.p, damping, epsilon, epsilon2, max_iter, push_div, &st))) break;
    st.h2d_ms = h2d;
    st.prep_ms = prep;
    if (hipMemcpy(scores, d_scores.p, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_pr_delta: download failed");
      rc = GDN_ERR_HIP;
      break;
    }
  } while (0);
  if (plan) gdn_pr_delta_plan_free(plan);
  if (go) gdn_graph_free(go);
  if (gi) gdn_graph_free(gi);
  if (rc == GDN_OK && stats) *stats = st;
  return rc;
}

}  // extern "C"
