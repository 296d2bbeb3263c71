// gdn_bc.hip -- betweenness centrality from one source (Brandes), SURVEY 8f rank 4.
//
// Reference path: BCSolver (src/bc/bc.h:37); OpenMP src/bc/omp_base.cc:55-105 = a forward BFS that counts shortest
// paths in ints (PBFS :16-53), the dependencies from the deepest level back (:78-93, float, per-source sum over the
// successors in CSR order), every score divided by the largest (:95-101).  CUDA: src/bc/linear_base.cu.
//
// Here: the level-synchronous forward phase runs on the load-balanced expansion of gdn_expand.hpp (64 frontier
// vertices per wave, long rows as work items); ALL levels stay in one array `order` (the reference's SlidingQueue),
// discovery = atomicCAS on depth, path counts = atomicAdd in (wrapping) 32-bit ints like the reference -- integer
// addition is order independent, so depths and path counts are bit-identical to it.  The successor bitmap of the
// reference (1 bit per edge) is not kept: successor(src -> dst) <=> depth[dst] == depth[src] + 1.
// The backward phase visits one level per launch: short rows are summed by ONE lane in CSR order with the
// reference's fp32 operations (no FMA contraction) -- the same bits as the reference for every vertex with fewer than
// BC_WAVE_ROW out-edges -- rows from BC_WAVE_ROW edges on by a whole wave, rows from BC_BLOCK_ROW on by a workgroup in a
// second launch (fixed reduction trees: deterministic, within 1e-6 of the sequential sum).
#include <string.h>

#include <vector>

#include "gdn_expand.hpp"

#define BC_WAVE_ROW 32     // rows at least this long: whole wave
#define BC_BLOCK_ROW 4096  // rows at least this long: one workgroup each (second launch)

struct BcCounters {  // device
  unsigned tail;       // entries of `order` (the next level is appended behind the current one)
  unsigned big_count;  // forward: big-row work items; backward: rows left to the workgroup kernel
  unsigned overflow;
  unsigned pad;
};

struct BcFwdVis {
  const vid_t *__restrict__ colidx;
  int32_t *__restrict__ depth;
  int32_t *__restrict__ pc;
  vid_t *__restrict__ order;
  BcCounters *cnt;
  unsigned cap;
  int32_t next_level;
  int32_t pc_src;  // path count of this lane's frontier vertex (big items: of the item's vertex, same in every lane)
  int big;
  GdnWlStage stage;
  __device__ __forceinline__ void begin_big(vid_t v) {
    big = 1;
    pc_src = pc[v];
  }
  __device__ __forceinline__ void edge(int owner, eoff_t k, bool valid) {
    const int32_t ps = big ? pc_src : __shfl(pc_src, owner, 64);
    bool claim = false;
    vid_t dst = 0;
    if (valid) {
      dst = __builtin_nontemporal_load(colidx + k);
      int32_t d = depth[dst];  // -1 can be stale (another XCD claimed it): the CAS decides
      if (d == -1) {
        const int32_t old = atomicCAS(&depth[dst], -1, next_level);
        claim = old == -1;
        d = claim ? next_level : old;
      }
      if (d == next_level) atomicAdd(&pc[dst], ps);  // src/bc/omp_base.cc:39-42
    }
    gdn_wl_push_staged(stage, order, &cnt->tail, cap, claim, dst, &cnt->overflow);
  }
  __device__ __forceinline__ void finish() { gdn_wl_flush(stage, order, &cnt->tail, cap, &cnt->overflow); }
};

__global__ void __launch_bounds__(GDN_BLOCK)
bc_fwd_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ level, unsigned nf, ExpBigList big, BcFwdVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  __shared__ vid_t s_stage[GDN_WAVES_PER_BLOCK][GDN_WL_STAGE];
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vid_t v = 0;
  vis.pc_src = 0;
  if (i < nf) {
    v = level[i];
    b = rowptr[v];
    e = rowptr[v + 1];
    vis.pc_src = vis.pc[v];
  }
  vis.big = 0;
  vis.stage.strip = s_stage[threadIdx.x >> 6];
  vis.stage.n = 0;
  gdn_expand_wave(b, e, v, big, vis, s_scan[threadIdx.x >> 6]);
  vis.finish();
}

__global__ void __launch_bounds__(GDN_BLOCK)
bc_fwd_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, BcFwdVis vis) {
  __shared__ vid_t s_stage[GDN_WAVES_PER_BLOCK][GDN_WL_STAGE];
  vis.stage.strip = s_stage[threadIdx.x >> 6];
  vis.stage.n = 0;
  vis.big = 1;
  vis.pc_src = 0;
  gdn_expand_big_items(rowptr, big, vis);
  vis.finish();
}

__global__ void bc_seed_kernel(int32_t source, int32_t *depth, int32_t *pc, vid_t *order, BcCounters *cnt) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    depth[source] = 0;
    pc[source] = 1;
    order[0] = source;
    cnt->tail = 1;
    cnt->big_count = 0;
    cnt->overflow = 0;
  }
}

// one term of src/bc/omp_base.cc:87-88, in the reference's operation order and without contraction
__device__ __forceinline__ float bc_term(float pcs, int32_t pcd, float delta_dst) {
  return __fmul_rn(__fdiv_rn(pcs, (float)pcd), __fadd_rn(1.0f, delta_dst));
}

// What the backward sweep reads of a successor, in ONE 16-byte record (one divergent access per edge instead of three):
// x = depth, y = path count, z = bits of delta, w unused
typedef int bc_i32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(GDN_BLOCK)
bc_pack_kernel(const int32_t *__restrict__ depth, const int32_t *__restrict__ pc, int32_t m, bc_i32x4 *__restrict__ rec) {
  const size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (size_t)m) rec[v] = bc_i32x4{depth[v], pc[v], 0, 0};
}

__device__ __forceinline__ float bc_edge_term(const bc_i32x4 *__restrict__ rec, vid_t dst, int32_t next_level, float pcs) {
  const bc_i32x4 r = rec[dst];
  return r.x == next_level ? bc_term(pcs, r.y, __int_as_float(r.z)) : 0.0f;
}

// backward step of one level: delta[src] = SUM over successors, scores[src] += delta[src].  A term of a non-successor
// is +0.0f, which leaves every partial sum unchanged (the sums are never -0.0f), so the loops carry no branch.
#define BC_UNR 4
__global__ void __launch_bounds__(GDN_BLOCK)
bc_back_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const vid_t *__restrict__ level, unsigned nf,
               bc_i32x4 *__restrict__ rec, float *__restrict__ scores, int32_t next_level, vid_t *__restrict__ big_rows,
               BcCounters *cnt, unsigned cap) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned lane = gdn_lane();
  eoff_t b = 0, e = 0;
  vid_t v = 0;
  float pcs = 0.0f;
  if (i < nf) {
    v = level[i];
    b = rowptr[v];
    e = rowptr[v + 1];
    pcs = (float)rec[v].y;
  }
  const eoff_t deg = e - b;
  float acc = 0.0f;
  // rows for the workgroup kernel
  const bool is_big = deg >= BC_BLOCK_ROW;
  gdn_wl_push(big_rows, &cnt->big_count, cap, is_big, v, &cnt->overflow);
  // medium rows: the whole wave, one row at a time; lane l sums edges l, l + 64, ... (BC_UNR of them in flight), then a
  // fixed shuffle tree
  unsigned long long mask = __ballot(deg >= BC_WAVE_ROW && !is_big);
  while (mask) {
    const int leader = __ffsll((long long)mask) - 1;
    mask &= mask - 1ull;
    const eoff_t bb = __shfl(b, leader, 64), ee = __shfl(e, leader, 64);
    const float ps = __shfl(pcs, leader, 64);
    float part = 0.0f;
    for (eoff_t k0 = bb + lane; k0 < ee; k0 += 64 * BC_UNR) {
      vid_t dst[BC_UNR];
      float t[BC_UNR];
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) dst[r] = k0 + 64 * r < ee ? colidx[k0 + 64 * r] : -1;
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) t[r] = dst[r] >= 0 ? bc_edge_term(rec, dst[r], next_level, ps) : 0.0f;
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) part = __fadd_rn(part, t[r]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part = __fadd_rn(part, __shfl_xor(part, o, 64));
    if ((int)lane == leader) acc = part;
  }
  // short rows: one lane, CSR order, the reference's arithmetic (the loads of BC_UNR edges in flight, added in order)
  if (i < nf && deg < BC_WAVE_ROW) {
    for (eoff_t k0 = b; k0 < e; k0 += BC_UNR) {
      vid_t dst[BC_UNR];
      float t[BC_UNR];
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) dst[r] = k0 + r < e ? colidx[k0 + r] : -1;
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) t[r] = dst[r] >= 0 ? bc_edge_term(rec, dst[r], next_level, pcs) : 0.0f;
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) acc = __fadd_rn(acc, t[r]);
    }
  }
  if (i < nf && !is_big) {
    rec[v].z = __float_as_int(acc);
    scores[v] = __fadd_rn(scores[v], acc);
  }
}

// rows from BC_BLOCK_ROW edges on: one 1024-thread workgroup each, BC_UNR edges per thread in flight
#define BC_BIG_THREADS 1024
__global__ void __launch_bounds__(BC_BIG_THREADS)
bc_back_big_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const vid_t *__restrict__ big_rows,
                   const BcCounters *__restrict__ cnt, bc_i32x4 *__restrict__ rec, float *__restrict__ scores, int32_t next_level,
                   unsigned cap) {
  __shared__ float s_red[BC_BIG_THREADS / 64];
  unsigned n = cnt->big_count;
  if (n > cap) n = cap;
  for (unsigned r0 = blockIdx.x; r0 < n; r0 += gridDim.x) {
    const vid_t v = big_rows[r0];
    const eoff_t b = rowptr[v], e = rowptr[v + 1];
    const float pcs = (float)rec[v].y;
    float part = 0.0f;
    for (eoff_t k0 = b + threadIdx.x; k0 < e; k0 += (eoff_t)BC_BIG_THREADS * BC_UNR) {
      vid_t dst[BC_UNR];
      float t[BC_UNR];
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) dst[r] = k0 + (eoff_t)BC_BIG_THREADS * r < e ? colidx[k0 + (eoff_t)BC_BIG_THREADS * r] : -1;
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) t[r] = dst[r] >= 0 ? bc_edge_term(rec, dst[r], next_level, pcs) : 0.0f;
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) part = __fadd_rn(part, t[r]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part = __fadd_rn(part, __shfl_xor(part, o, 64));
    __syncthreads();
    if (gdn_lane() == 0) s_red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.0f;
      for (int w = 0; w < BC_BIG_THREADS / 64; w++) t = __fadd_rn(t, s_red[w]);
      rec[v].z = __float_as_int(t);
      scores[v] = __fadd_rn(scores[v], t);
    }
  }
}

// max of non-negative floats as their bit patterns (NaN inputs do not occur: scores start finite and grow by finite terms)
__global__ void __launch_bounds__(GDN_BLOCK)
bc_max_kernel(const float *__restrict__ scores, int32_t m, unsigned *__restrict__ out) {
  float mx = 0.0f;
  for (size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < (size_t)m; i += (size_t)gridDim.x * GDN_BLOCK)
    mx = fmaxf(mx, scores[i]);  // max(biggest, score) with biggest starting at 0 (src/bc/omp_base.cc:96-99)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (gdn_lane() == 0 && mx > 0.0f) atomicMax(out, __float_as_uint(mx));
}

__global__ void __launch_bounds__(GDN_BLOCK)
bc_normalize_kernel(float *__restrict__ scores, int32_t m, const unsigned *__restrict__ mx) {
  const float big = __uint_as_float(*mx);
  const size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < (size_t)m) scores[i] = __fdiv_rn(scores[i], big);  // 0/0 = NaN when nothing lies between, like the reference
}

int gdn_reached_edges(const gdn_graph *g, const int32_t *d_dist, int32_t unreached, uint64_t *out);

extern "C" {

int gdn_bc_dev(const gdn_graph *g, int32_t source, float *d_scores, gdn_stats *stats) {
  GDN_REQUIRE(g != nullptr && d_scores != nullptr, "graph / d_scores");
  GDN_REQUIRE(source >= 0 && source < g->m, "source out of range");
  const int32_t m = g->m;
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  DevBuf<int32_t> depth, pc;
  DevBuf<bc_i32x4> rec;
  DevBuf<vid_t> order, big_rows;
  DevBuf<unsigned long long> bigitems;
  DevBuf<BcCounters> cnt;
  DevBuf<unsigned> mx;
  const uint64_t bigcap64 = g->nnz / EXP_CHUNK + (uint64_t)m / 64 + 1024;
  const unsigned bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
  const unsigned rowcap = (unsigned)(g->nnz / BC_BLOCK_ROW + 16);
  GDN_TRY(depth.alloc((size_t)m));
  GDN_TRY(pc.alloc((size_t)m));
  GDN_TRY(rec.alloc((size_t)m));
  GDN_TRY(order.alloc((size_t)m));
  GDN_TRY(big_rows.alloc(rowcap));
  GDN_TRY(bigitems.alloc(bigcap));
  GDN_TRY(cnt.alloc(1));
  GDN_TRY(mx.alloc(1));
  HostTimer tsolve;
  // ---- timed region == src/bc/omp_base.cc:66-102 (t.Start .. t.Stop), the per-iteration vectors included
  tsolve.start();
  GDN_TRY(gdn_fill_i32(depth.p, -1, (size_t)m, 0));
  GDN_HIP(hipMemsetAsync(pc.p, 0, (size_t)m * 4, 0));
  GDN_HIP(hipMemsetAsync(mx.p, 0, 4, 0));
  hipLaunchKernelGGL(bc_seed_kernel, dim3(1), dim3(64), 0, 0, source, depth.p, pc.p, order.p, cnt.p);
  // forward: level d = order[lp[d] .. lp[d+1])
  std::vector<unsigned> lp;
  lp.push_back(0);
  lp.push_back(1);
  ExpBigList big;
  big.items = bigitems.p;
  big.capacity = bigcap;
  big.count = &cnt.p->big_count;
  big.overflow = &cnt.p->overflow;
  BcCounters h;
  for (int32_t level = 0;; level++) {
    const unsigned l0 = lp[(size_t)level], nf = lp[(size_t)level + 1] - l0;
    if (nf == 0) break;
    BcFwdVis vis;
    vis.colidx = g->colidx;
    vis.depth = depth.p;
    vis.pc = pc.p;
    vis.order = order.p;
    vis.cnt = cnt.p;
    vis.cap = (unsigned)m;
    vis.next_level = level + 1;
    vis.pc_src = 0;
    vis.big = 0;
    hipLaunchKernelGGL(bc_fwd_kernel, dim3(gdn_nblocks(nf)), dim3(GDN_BLOCK), 0, 0, g->rowptr, order.p + l0, nf, big, vis);
    hipLaunchKernelGGL(bc_fwd_big_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
    GDN_HIP(hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost));
    if (h.overflow) {
      gdn_set_error("gdn_bc: device worklist overflow");
      return GDN_ERR_OVERFLOW;
    }
    lp.push_back(h.tail);
    GDN_HIP(hipMemsetAsync(&cnt.p->big_count, 0, sizeof(unsigned), 0));
  }
  const int32_t nlev = (int32_t)lp.size() - 2;  // non-empty levels 0 .. nlev-1
  // backward: the deepest level has no successors (its deltas stay 0, like the reference's first sweep)
  hipLaunchKernelGGL(bc_pack_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, depth.p, pc.p, m, rec.p);
  for (int32_t d = nlev - 2; d >= 0; d--) {
    const unsigned l0 = lp[(size_t)d], nf = lp[(size_t)d + 1] - l0;
    hipLaunchKernelGGL(bc_back_kernel, dim3(gdn_nblocks(nf)), dim3(GDN_BLOCK), 0, 0, g->rowptr, g->colidx, order.p + l0, nf,
                       rec.p, d_scores, d + 1, big_rows.p, cnt.p, rowcap);
    hipLaunchKernelGGL(bc_back_big_kernel, dim3(512), dim3(BC_BIG_THREADS), 0, 0, g->rowptr, g->colidx, big_rows.p, cnt.p, rec.p,
                       d_scores, d + 1, rowcap);
    GDN_HIP(hipMemsetAsync(&cnt.p->big_count, 0, sizeof(unsigned), 0));
  }
  hipLaunchKernelGGL(bc_max_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, d_scores, m, mx.p);
  hipLaunchKernelGGL(bc_normalize_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, d_scores, m, mx.p);
  GDN_HIP(hipGetLastError());
  GDN_HIP(hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost));
  if (h.overflow) {
    gdn_set_error("gdn_bc: device worklist overflow");
    return GDN_ERR_OVERFLOW;
  }
  st.solve_ms = tsolve.stop_ms();
  st.iterations = nlev;
  uint64_t te = 0;
  GDN_TRY(gdn_reached_edges(g, depth.p, -1, &te));
  st.edges_traversed = 2 * te;  // every out-edge of a reached vertex once forward, once backward
  if (stats) *stats = st;
  return GDN_OK;
}

// Host API: one call == BCSolver(g, source, scores) (src/bc/main.cc:22).
int gdn_bc(int32_t m, uint64_t nnz, const uint64_t *rowptr, const int32_t *colidx, int32_t source, float *scores,
           gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && rowptr && scores && (colidx || nnz == 0), "null argument");
  GDN_REQUIRE(source >= 0 && source < m, "source out of range");
  GDN_TRY(gdn_require_device());
  HostTimer th2d;
  th2d.start();
  gdn_graph *g = nullptr;
  GDN_TRY(gdn_graph_upload(m, nnz, rowptr, colidx, &g));
  DevBuf<float> d_scores;
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  int rc = GDN_OK;
  do {
    if ((rc = d_scores.alloc(m))) break;
    if (hipMemcpy(d_scores.p, scores, (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess) {
      gdn_set_error("gdn_bc: upload failed");
      rc = GDN_ERR_HIP;
      break;
    }
    const double h2d = th2d.stop_ms();
    if ((rc = gdn_bc_dev(g, source, d_scores.p, &st))) break;
    st.h2d_ms = h2d;
    if (hipMemcpy(scores, d_scores.p, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_bc: download failed");
      rc = GDN_ERR_HIP;
    }
  } while (0);
  gdn_graph_free(g);
  if (stats) *stats = st;
  return rc;
}

}  // extern "C"
