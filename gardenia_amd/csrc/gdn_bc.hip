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

#include <algorithm>
#include <vector>

#include "gdn_expand.hpp"
#include "gdn_pb.hpp"

#define BC_WAVE_ROW 32     // rows at least this long: whole wave
#define BC_BLOCK_ROW 4096  // rows at least this long: one workgroup each (second launch)

struct BcCounters {  // device; the two hot counters on 128-byte lines of their own (atomics on one line serialise, gdn_sssp.hip)
  alignas(128) unsigned tail;       // entries of `order` (the next level is appended behind the current one)
  alignas(128) unsigned big_count;  // forward: big-row work items; backward: rows left to the workgroup kernel
  alignas(128) unsigned overflow;
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
  // batched levels (bc_fwd_lvl_kernel): the next level is appended at q[*qcount ...] instead of order[cnt->tail ...]
  vid_t *q = nullptr;
  unsigned *qcount = nullptr;
  unsigned qcap = 0;
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
    gdn_wl_push_staged(stage, q ? q : order, q ? qcount : &cnt->tail, q ? qcap : cap, claim, dst, &cnt->overflow);
  }
  __device__ __forceinline__ void finish() { gdn_wl_flush(stage, q ? q : order, q ? qcount : &cnt->tail, q ? qcap : cap, &cnt->overflow); }
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

// ---- forward levels queued in BATCHES (mid-size levels of a high-diameter graph: a lattice level of a few thousand
// vertices is two launches and a blocking read of its size, ~25 us): the level kernels take the bounds of their level
// from a device array the previous level's closing kernel has written, so the host queues several levels before it
// reads anything back; a level behind the last one finds itself empty.  Level j of the batch = order[tails[j], tails[j+1]).
__global__ void __launch_bounds__(GDN_BLOCK)
bc_fwd_lvl_kernel(const eoff_t *__restrict__ rowptr, vid_t *order, unsigned *counts /* [0] start of level 0 of the batch, [1 + j] size of level j */,
                  int j, ExpBigList big, BcFwdVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  __shared__ vid_t s_stage[GDN_WAVES_PER_BLOCK][GDN_WL_STAGE];
  unsigned l0 = counts[0];
  for (int i = 0; i < j; i++) l0 += counts[1 + i];  // the levels in front of this one are complete
  const unsigned nf = counts[1 + j];
  // the vertices this level discovers go behind it, counted in counts[2 + j] (zero when the batch was queued): no closing
  // kernel between two levels
  vis.q = order + l0 + nf;
  vis.qcount = counts + 2 + j;
  vis.qcap = vis.cap - (l0 + nf);
  vis.stage.strip = s_stage[threadIdx.x >> 6];
  vis.stage.n = 0;
  for (unsigned base = blockIdx.x * GDN_BLOCK; base < nf; base += gridDim.x * GDN_BLOCK) {
    const unsigned i = base + threadIdx.x;
    eoff_t b = 0, e = 0;
    vid_t v = 0;
    vis.pc_src = 0;
    if (i < nf) {
      v = order[l0 + i];
      b = rowptr[v];
      e = rowptr[v + 1];
      vis.pc_src = vis.pc[v];
    }
    vis.big = 0;
    gdn_expand_wave(b, e, v, big, vis, s_scan[threadIdx.x >> 6]);
  }
  vis.finish();
}

// big-row work items of a batched level (same queue indirection), and the reset of their counter for the next level
__global__ void __launch_bounds__(GDN_BLOCK)
bc_fwd_lvl_big_kernel(const eoff_t *__restrict__ rowptr, unsigned *counts, int j, ExpBigList big, BcFwdVis vis) {
  __shared__ vid_t s_stage[GDN_WAVES_PER_BLOCK][GDN_WL_STAGE];
  unsigned l0 = counts[0];
  for (int i = 0; i < j; i++) l0 += counts[1 + i];
  const unsigned nf = counts[1 + j];
  vis.q = vis.order + l0 + nf;
  vis.qcount = counts + 2 + j;
  vis.qcap = vis.cap - (l0 + nf);
  vis.stage.strip = s_stage[threadIdx.x >> 6];
  vis.stage.n = 0;
  vis.big = 1;
  vis.pc_src = 0;
  gdn_expand_big_items(rowptr, big, vis);
  vis.finish();
}

__global__ void bc_fwd_lvl_end_kernel(BcCounters *cnt) {
  if (threadIdx.x == 0 && blockIdx.x == 0) cnt->big_count = 0;
}

// ---- fused light levels of the forward phase (cf. bfs_td_small_kernel, gdn_bfs.hip): ONE workgroup runs consecutive
// levels while the frontier stays small -- a lane per short row, a wave per longer one, the level boundary is a
// __syncthreads() -- and
// reports the tail of `order` after every level.  A level costs ~5 us here instead of two launches and a blocking read
// back: every level of a high-diameter graph.  Path counts are integer atomics: the same bits in any order.
#define BC_SMALL_THREADS 1024
#define BC_SMALL_MAX_LEVELS 2048
#define BC_SMALL_LANE_ROW 32  // rows up to this long: one lane each
#define BC_SMALL_UNR 4
struct BcSmallOut {
  unsigned levels;    // levels expanded here (0: the frontier handed in was too heavy, nothing was touched)
  unsigned overflow;
  unsigned tails[BC_SMALL_MAX_LEVELS];  // entries of `order` after each of them
};

__global__ void __launch_bounds__(BC_SMALL_THREADS)
bc_fwd_small_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, int32_t *__restrict__ depth,
                    int32_t *pc, vid_t *order, unsigned l0, unsigned nf, unsigned cap, int32_t level, unsigned max_nf,
                    unsigned long long max_scout, BcCounters *cnt, BcSmallOut *__restrict__ out) {
  __shared__ unsigned s_tail, s_over, s_nlong;
  __shared__ unsigned long long s_scout;
  __shared__ vid_t s_long[BC_SMALL_THREADS];
  // the level just discovered, with its rows' bounds (read for the scout count anyway): vertex and row pointers of the
  // next level's lanes come from LDS instead of two dependent global reads
  __shared__ vid_t s_qv[2][BC_SMALL_THREADS];
  __shared__ eoff_t s_qb[2][BC_SMALL_THREADS], s_qe[2][BC_SMALL_THREADS];
  const unsigned lane = gdn_lane(), wave = threadIdx.x >> 6, nwaves = BC_SMALL_THREADS / 64;
  unsigned levels = 0, tail = l0 + nf, par = 0;
  // out-edges of the frontier handed in: a heavy one goes back to the host untouched
  if (threadIdx.x == 0) {
    s_scout = 0ull;
    s_over = 0u;
    s_nlong = 0u;
  }
  __syncthreads();
  {
    unsigned long long sc = 0;
    for (unsigned i = threadIdx.x; i < nf; i += BC_SMALL_THREADS) {
      const vid_t v = __hip_atomic_load(order + l0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sc += rowptr[v + 1] - rowptr[v];
    }
    sc = gdn_wave_sum(sc);
    if (lane == 0 && sc) atomicAdd(&s_scout, sc);
  }
  __syncthreads();
  unsigned long long scout_cur = s_scout;
  __syncthreads();
  while (nf > 0 && nf <= max_nf && scout_cur <= max_scout && levels < BC_SMALL_MAX_LEVELS && !s_over) {
    if (threadIdx.x == 0) {
      s_tail = tail;
      s_scout = 0ull;
    }
    __syncthreads();
    unsigned long long scout = 0;
    // one edge: discovery = CAS on depth, path counts = integer atomics (src/bc/omp_base.cc:39-42); a claimed vertex goes
    // to the tail of `order`, one LDS atomic per wave step
    auto visit = [&](bool valid, vid_t dst, int32_t ps) {
      bool claim = false;
      if (valid) {  // no read in front of the CAS: every memory round trip of a light level is on its critical path
        const int32_t old = atomicCAS(&depth[dst], -1, level + 1);
        claim = old == -1;
        if (claim || old == level + 1) atomicAdd(&pc[dst], ps);
      }
      const unsigned long long mask = __ballot(claim);
      if (mask) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&s_tail, (unsigned)__popcll(mask));
        base = __shfl(base, 0, 64);
        if (claim) {
          const eoff_t db = rowptr[dst], de = rowptr[dst + 1];
          scout += de - db;
          const unsigned pos = base + (unsigned)__popcll(mask & gdn_lanemask_lt());
          if (pos < cap) __hip_atomic_store(order + pos, dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else s_over = 1u;
          if (pos - tail < BC_SMALL_THREADS) {  // the next level finds its first entries (all, when it is small) here
            s_qv[par ^ 1u][pos - tail] = dst;
            s_qb[par ^ 1u][pos - tail] = db;
            s_qe[par ^ 1u][pos - tail] = de;
          }
        }
      }
    };
    const bool from_lds = levels > 0 && nf <= BC_SMALL_THREADS;
    for (unsigned c0 = 0; c0 < nf; c0 += BC_SMALL_THREADS) {
      // values other waves of this workgroup wrote (queue entries, path counts summed by atomics): device-scope loads
      eoff_t b = 0, e = 0;
      int32_t ps = 0;
      if (c0 + threadIdx.x < nf) {
        vid_t v;
        if (from_lds) {
          v = s_qv[par][threadIdx.x];
          b = s_qb[par][threadIdx.x];
          e = s_qe[par][threadIdx.x];
        } else {
          v = __hip_atomic_load(order + l0 + c0 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          b = rowptr[v];
          e = rowptr[v + 1];
        }
        ps = __hip_atomic_load(pc + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e - b > BC_SMALL_LANE_ROW) {
          s_long[atomicAdd(&s_nlong, 1u)] = v;
          e = b;
        }
      }
      // short rows: a lane each, BC_SMALL_UNR edges of it in flight
      for (eoff_t k0 = b; __any(k0 < e); k0 += BC_SMALL_UNR) {
        vid_t dst[BC_SMALL_UNR];
#pragma unroll
        for (int r = 0; r < BC_SMALL_UNR; r++) dst[r] = k0 + r < e ? colidx[k0 + r] : -1;
#pragma unroll
        for (int r = 0; r < BC_SMALL_UNR; r++) visit(dst[r] >= 0, dst[r], ps);
      }
      __syncthreads();
      // longer rows: a wave each
      const unsigned nlong = s_nlong;
      for (unsigned i = wave; i < nlong; i += nwaves) {
        const vid_t v = s_long[i];
        const int32_t pv = __hip_atomic_load(pc + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const eoff_t bb = rowptr[v], ee = rowptr[v + 1];
        for (eoff_t k0 = bb; k0 < ee; k0 += 64) {
          const eoff_t k = k0 + lane;
          visit(k < ee, k < ee ? colidx[k] : 0, pv);
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) s_nlong = 0u;
      if (c0 + BC_SMALL_THREADS < nf) __syncthreads();
    }
    scout = gdn_wave_sum(scout);
    if (lane == 0 && scout) atomicAdd(&s_scout, scout);
    gdn_wg_level_sync();
    const unsigned new_tail = s_tail;
    scout_cur = s_scout;
    if (threadIdx.x == 0) out->tails[levels] = new_tail;
    l0 = tail;
    nf = new_tail - tail;
    tail = new_tail;
    level++;
    levels++;
    par ^= 1u;
    __syncthreads();  // everybody has read s_tail / s_scout before the next level resets them
  }
  if (threadIdx.x == 0) {
    out->levels = levels;
    out->overflow = s_over;
    if (levels) cnt->tail = tail < cap ? tail : cap;
    if (s_over) cnt->overflow = 1u;
  }
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
  return gdn_fmul(__fdiv_rn(pcs, (float)pcd), gdn_fadd(1.0f, delta_dst));
}

// What the backward sweep reads of a successor, in ONE 16-byte record (one divergent access per edge instead of three):
// x = depth, y = path count, z = bits of delta, w unused
typedef int bc_i32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(GDN_BLOCK)
bc_pack_kernel(const int32_t *__restrict__ depth, const int32_t *__restrict__ pc, int32_t m, bc_i32x4 *__restrict__ rec) {
  const size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (size_t)m) rec[v] = bc_i32x4{depth[v], pc[v], 0, 0};
}

// FUSED: the deltas were written one level ago by other waves of the SAME launch (bc_back_small_kernel) -- device-scope
// loads, past this CU's vector cache
template <bool FUSED>
__device__ __forceinline__ float bc_edge_term(const bc_i32x4 *rec, vid_t dst, int32_t next_level, float pcs) {
  bc_i32x4 r;
  if (FUSED) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(rec + dst);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r = bc_i32x4{(int)(unsigned)lo, (int)(unsigned)(lo >> 32), (int)(unsigned)hi, 0};
  } else {
    r = rec[dst];
  }
  return r.x == next_level ? bc_term(pcs, r.y, __int_as_float(r.z)) : 0.0f;
}

// backward step of one level: delta[src] = SUM over successors, scores[src] += delta[src].  A term of a non-successor
// is +0.0f, which leaves every partial sum unchanged (the sums are never -0.0f), so the loops carry no branch.
// The three row classes and their summation orders are the same in every kernel that calls these two functions: a
// vertex's delta does not depend on which kernel ran its level.
#define BC_UNR 4
#define BC_BIG_THREADS 1024

// rows below BC_BLOCK_ROW edges, one per lane (b == e in idle lanes); returns the lane's row sum
template <bool FUSED>
__device__ __forceinline__ float bc_back_wave_rows(const vid_t *__restrict__ colidx, const bc_i32x4 *rec, int32_t next_level, eoff_t b,
                                                   eoff_t e, float pcs, bool is_big) {
  const unsigned lane = gdn_lane();
  const eoff_t deg = e - b;
  float acc = 0.0f;
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
      for (int r = 0; r < BC_UNR; r++) t[r] = dst[r] >= 0 ? bc_edge_term<FUSED>(rec, dst[r], next_level, ps) : 0.0f;
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) part = gdn_fadd(part, t[r]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part = gdn_fadd(part, __shfl_xor(part, o, 64));
    if ((int)lane == leader) acc = part;
  }
  // short rows: one lane, CSR order, the reference's arithmetic (the loads of BC_UNR edges in flight, added in order)
  if (deg < BC_WAVE_ROW) {
    for (eoff_t k0 = b; k0 < e; k0 += BC_UNR) {
      vid_t dst[BC_UNR];
      float t[BC_UNR];
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) dst[r] = k0 + r < e ? colidx[k0 + r] : -1;
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) t[r] = dst[r] >= 0 ? bc_edge_term<FUSED>(rec, dst[r], next_level, pcs) : 0.0f;
#pragma unroll
      for (int r = 0; r < BC_UNR; r++) acc = gdn_fadd(acc, t[r]);
    }
  }
  return acc;
}

// one row from BC_BLOCK_ROW edges on by a whole 1024-thread workgroup, BC_UNR edges per thread in flight; the sum is
// returned in thread 0 (s_red: BC_BIG_THREADS / 64 floats; contains __syncthreads)
template <bool FUSED>
__device__ __forceinline__ float bc_back_block_row(const vid_t *__restrict__ colidx, const bc_i32x4 *rec, int32_t next_level, eoff_t b,
                                                   eoff_t e, float pcs, float *s_red) {
  float part = 0.0f;
  for (eoff_t k0 = b + threadIdx.x; k0 < e; k0 += (eoff_t)BC_BIG_THREADS * BC_UNR) {
    vid_t dst[BC_UNR];
    float t[BC_UNR];
#pragma unroll
    for (int r = 0; r < BC_UNR; r++) dst[r] = k0 + (eoff_t)BC_BIG_THREADS * r < e ? colidx[k0 + (eoff_t)BC_BIG_THREADS * r] : -1;
#pragma unroll
    for (int r = 0; r < BC_UNR; r++) t[r] = dst[r] >= 0 ? bc_edge_term<FUSED>(rec, dst[r], next_level, pcs) : 0.0f;
#pragma unroll
    for (int r = 0; r < BC_UNR; r++) part = gdn_fadd(part, t[r]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part = gdn_fadd(part, __shfl_xor(part, o, 64));
  __syncthreads();
  if (gdn_lane() == 0) s_red[threadIdx.x >> 6] = part;
  __syncthreads();
  float t = 0.0f;
  if (threadIdx.x == 0)
    for (int w = 0; w < BC_BIG_THREADS / 64; w++) t = gdn_fadd(t, s_red[w]);
  return t;
}

__global__ void __launch_bounds__(GDN_BLOCK)
bc_back_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const vid_t *__restrict__ level, unsigned nf,
               bc_i32x4 *__restrict__ rec, float *__restrict__ scores, int32_t next_level, vid_t *__restrict__ big_rows,
               BcCounters *cnt, unsigned cap) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vid_t v = 0;
  float pcs = 0.0f;
  if (i < nf) {
    v = level[i];
    b = rowptr[v];
    e = rowptr[v + 1];
    pcs = (float)rec[v].y;
  }
  // rows for the workgroup kernel
  const bool is_big = e - b >= BC_BLOCK_ROW;
  gdn_wl_push(big_rows, &cnt->big_count, cap, is_big, v, &cnt->overflow);
  const float acc = bc_back_wave_rows<false>(colidx, rec, next_level, b, e, pcs, is_big);
  if (i < nf && !is_big) {
    rec[v].z = __float_as_int(acc);
    scores[v] = gdn_fadd(scores[v], acc);
  }
}

__global__ void __launch_bounds__(BC_BIG_THREADS)
bc_back_big_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const vid_t *__restrict__ big_rows,
                   const BcCounters *__restrict__ cnt, bc_i32x4 *__restrict__ rec, float *__restrict__ scores, int32_t next_level,
                   unsigned cap) {
  __shared__ float s_red[BC_BIG_THREADS / 64];
  unsigned n = cnt->big_count;
  if (n > cap) n = cap;
  for (unsigned r0 = blockIdx.x; r0 < n; r0 += gridDim.x) {
    const vid_t v = big_rows[r0];
    const float t = bc_back_block_row<false>(colidx, rec, next_level, rowptr[v], rowptr[v + 1], (float)rec[v].y, s_red);
    if (threadIdx.x == 0) {
      rec[v].z = __float_as_int(t);
      scores[v] = gdn_fadd(scores[v], t);
    }
  }
}

// ---- fused light levels of the backward phase (the mirror of bc_fwd_small_kernel): ONE workgroup walks consecutive
// levels from `d` down while a level has at most max_nf (<= 1024) vertices and max_scout out-edges -- a lane per short row,
// a wave per medium row, the workgroup per long row, exactly as the per-level kernels sum them -- and reports the first
// level it did not take (-1: none left).  lp[d] .. lp[d+1] = the level's entries of `order`.
__global__ void __launch_bounds__(BC_SMALL_THREADS)
bc_back_small_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const vid_t *__restrict__ order,
                     const unsigned *__restrict__ lp, int32_t d, unsigned max_nf, unsigned long long max_scout, bc_i32x4 *rec,
                     float *__restrict__ scores, int32_t *__restrict__ out_next) {
  static_assert(BC_SMALL_THREADS == BC_BIG_THREADS, "long rows are summed by BC_BIG_THREADS threads in every kernel");
  __shared__ float s_red[BC_BIG_THREADS / 64];
  __shared__ unsigned long long s_scout[2];
  __shared__ unsigned s_nbig[2];
  __shared__ vid_t s_big[BC_SMALL_THREADS];
  if (threadIdx.x < 2) {
    s_scout[threadIdx.x] = 0ull;
    s_nbig[threadIdx.x] = 0u;
  }
  __syncthreads();
  // a level's vertices, row bounds and path counts are final: the NEXT level's are loaded while this one is summed, off
  // the critical path of dependent reads
  struct Lvl {
    unsigned nf;
    vid_t v;
    eoff_t b, e;
    float pcs;
  };
  auto load_level = [&](int32_t dd) {
    Lvl L{0xFFFFFFFFu, 0, 0, 0, 0.0f};
    if (dd < 0) return L;
    const unsigned l0 = lp[dd];
    L.nf = lp[dd + 1] - l0;
    if (L.nf <= max_nf && threadIdx.x < L.nf) {
      L.v = order[l0 + threadIdx.x];
      L.b = rowptr[L.v];
      L.e = rowptr[L.v + 1];
      L.pcs = (float)rec[L.v].y;  // depth and path count of a record are final since bc_pack_kernel
    }
    return L;
  };
  Lvl nxt = load_level(d);
  for (unsigned par = 0; d >= 0; d--, par ^= 1u) {
    const Lvl cur = nxt;
    const unsigned nf = cur.nf;
    if (nf > max_nf) break;
    nxt = load_level(d - 1);
    const vid_t v = cur.v;
    const eoff_t b = cur.b, e = cur.e;
    const float pcs = cur.pcs;
    const unsigned long long sc = gdn_wave_sum((unsigned long long)(e - b));
    if (gdn_lane() == 0 && sc) atomicAdd(&s_scout[par], sc);
    const bool is_big = e - b >= BC_BLOCK_ROW;
    if (is_big) s_big[atomicAdd(&s_nbig[par], 1u)] = v;
    __syncthreads();
    if (s_scout[par] > max_scout) break;  // nothing of this level has been written yet
    if (threadIdx.x == 0) {                // the other parity's cells: read by everybody one barrier ago at the latest
      s_scout[par ^ 1u] = 0ull;
      s_nbig[par ^ 1u] = 0u;
    }
    const float acc = bc_back_wave_rows<true>(colidx, rec, d + 1, b, e, pcs, is_big);
    if (threadIdx.x < nf && !is_big) {
      __hip_atomic_store(reinterpret_cast<int *>(rec + v) + 2, __float_as_int(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      scores[v] = gdn_fadd(scores[v], acc);
    }
    const unsigned nbig = s_nbig[par];
    for (unsigned r0 = 0; r0 < nbig; r0++) {
      const vid_t bv = s_big[r0];
      const float t = bc_back_block_row<true>(colidx, rec, d + 1, rowptr[bv], rowptr[bv + 1], (float)rec[bv].y, s_red);
      if (threadIdx.x == 0) {
        __hip_atomic_store(reinterpret_cast<int *>(rec + bv) + 2, __float_as_int(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        scores[bv] = gdn_fadd(scores[bv], t);
      }
    }
    gdn_wg_level_sync();
  }
  if (threadIdx.x == 0) *out_next = d;
}

// max of non-negative floats as their bit patterns (NaN inputs do not occur: scores start finite and grow by finite terms)
__global__ void __launch_bounds__(GDN_BLOCK)
bc_max_kernel(const float *__restrict__ scores, int32_t m, unsigned *__restrict__ out) {
  float mx = 0.0f;
  for (size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < (size_t)m; i += (size_t)gridDim.x * GDN_BLOCK)
    mx = fmaxf(mx, scores[i]);  // max(biggest, score) with biggest starting at 0 (src/bc/omp_base.cc:96-99)
  // (non-negative floats order like their bits; one atomic per workgroup -- per wave, 8192 of them on one word took 0.1 ms)
  gdn_block_max_u32(mx > 0.0f ? __float_as_uint(mx) : 0u, out);
}

__global__ void __launch_bounds__(GDN_BLOCK)
bc_normalize_kernel(float *__restrict__ scores, int32_t m, const unsigned *__restrict__ mx) {
  const float big = __uint_as_float(*mx);
  const size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < (size_t)m) scores[i] = __fdiv_rn(scores[i], big);  // 0/0 = NaN when nothing lies between, like the reference
}

int gdn_reached_edges(const gdn_graph *g, const int32_t *d_dist, int32_t unreached, uint64_t *out);

extern "C" int gdn_bc_dev(const gdn_graph *g, int32_t source, float *d_scores, gdn_stats *stats);

// ------------------------------------------------------------------------------------------
// Resident plan (gdn_bc_plan_*): depths from the BFS plan (direction-optimising + dense sweeps, gdn_bfs.hip), and the
// two HEAVY levels of an R-MAT-like graph -- 1.5 G random atomics forward, 1.5 G record gathers backward in the path
// above -- as PROPAGATION-BLOCKED sweeps on the PageRank layout machinery (gdn_pb.hpp):
//   forward  level d -> d+1: pc[v] = SUM over in-neighbours u at depth d of pc[u]   (in-CSR layout, RAW 32-bit values,
//            integer LDS accumulation: the low 32 bits of the sum = the reference's wrapping int arithmetic, exact)
//   backward level d: delta[u] = pc[u] * SUM over out-neighbours v at depth d+1 of (1 + delta[v]) / pc[v]
//            (out-CSR layout; the values are scaled by a power of two into [0, 2^-21] so that a row sum stays below 1,
//            encoded like PageRank's contributions and accumulated in 2^-62 fixed point: deterministic, but rounded
//            differently from the reference's sequential fp32 sum -- inside its verifier's tolerance, not bit-equal)
// Light levels run vertex-parallel over all vertices with a depth filter (no per-level queues are kept).
// ------------------------------------------------------------------------------------------
#define BC_MAX_LEVELS 64

struct BcPcOp {  // epilogue of the forward sweep: rows at depth next_level take the summed path count
  const int32_t *__restrict__ depth;
  int32_t *__restrict__ pc;
  int32_t next_level;
  bool vec_ok;
  __device__ __forceinline__ unsigned long long to_fixed(float v, unsigned &) const { return (unsigned long long)__float_as_uint(v); }
  __device__ __forceinline__ float from_fixed(unsigned long long a, unsigned &) const { return __uint_as_float((unsigned)a); }
  struct Pre {
    int32_t d;
  };
  __device__ __forceinline__ Pre pre(int32_t row) const { return Pre{depth[row]}; }
  __device__ __forceinline__ double fin(int32_t row, float sum, const Pre &p) const {
    if (p.d == next_level) pc[row] = (int32_t)__float_as_uint(sum);
    return 0.0;
  }
  struct Pre4 {
    int32_t d;
  };
  __device__ __forceinline__ Pre4 pre4(int32_t) const { return Pre4{0}; }
  __device__ __forceinline__ double fin4(int32_t, const float (&)[4], const Pre4 &) const { return 0.0; }
};

struct BcBackOp {  // epilogue of the backward sweep: rows at depth `level` get delta = pc * sum * unscale
  bc_i32x4 *__restrict__ rec;
  float *__restrict__ scores;
  int32_t level;
  float unscale;
  bool vec_ok;
  __device__ __forceinline__ unsigned long long to_fixed(float v, unsigned &) const { return pb_decode(__float_as_uint(v)); }
  __device__ __forceinline__ float from_fixed(unsigned long long a, unsigned &bad) const {
    if (a >> 63) bad = 1u;
    return ldexpf((float)a, -PB_FIX_SHIFT);
  }
  struct Pre {
    bc_i32x4 r;
  };
  __device__ __forceinline__ Pre pre(int32_t row) const { return Pre{rec[row]}; }
  __device__ __forceinline__ double fin(int32_t row, float sum, const Pre &p) const {
    if (p.r.x == level) {
      const float dl = gdn_fmul((float)p.r.y, gdn_fmul(sum, unscale));
      rec[row].z = __float_as_int(dl);
      scores[row] = gdn_fadd(scores[row], dl);
    }
    return 0.0;
  }
  struct Pre4 {
    int32_t d;
  };
  __device__ __forceinline__ Pre4 pre4(int32_t) const { return Pre4{0}; }
  __device__ __forceinline__ double fin4(int32_t, const float (&)[4], const Pre4 &) const { return 0.0; }
};

struct BcLevelStats {
  unsigned long long edges[BC_MAX_LEVELS];  // out-edges of the vertices of a level
  unsigned long long count[BC_MAX_LEVELS];
  unsigned deep;                            // a vertex at depth >= BC_MAX_LEVELS exists
};

__global__ void __launch_bounds__(GDN_BLOCK)
bc_level_stats_kernel(const int32_t *__restrict__ depth, const eoff_t *__restrict__ rowptr, int32_t m, int32_t unreached,
                      BcLevelStats *__restrict__ out) {
  __shared__ unsigned long long s_e[BC_MAX_LEVELS], s_c[BC_MAX_LEVELS];
  if (threadIdx.x < BC_MAX_LEVELS) s_e[threadIdx.x] = s_c[threadIdx.x] = 0ull;
  __syncthreads();
  for (size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; v < (size_t)m; v += (size_t)gridDim.x * GDN_BLOCK) {
    const int32_t d = depth[v];
    if (d == unreached) continue;
    if (d >= BC_MAX_LEVELS) {
      out->deep = 1u;
      continue;
    }
    atomicAdd(&s_e[d], rowptr[v + 1] - rowptr[v]);
    atomicAdd(&s_c[d], 1ull);
  }
  __syncthreads();
  if (threadIdx.x < BC_MAX_LEVELS && s_c[threadIdx.x]) {
    atomicAdd(&out->edges[threadIdx.x], s_e[threadIdx.x]);
    atomicAdd(&out->count[threadIdx.x], s_c[threadIdx.x]);
  }
}

// light forward level: every vertex at depth `level` pushes its path count along its out-edges to the vertices one deeper
struct BcPushVis {
  const vid_t *__restrict__ colidx;
  const int32_t *__restrict__ depth;
  int32_t *__restrict__ pc;
  int32_t next_level;
  int32_t pc_src;
  int big;
  __device__ __forceinline__ void begin_big(vid_t v) {
    big = 1;
    pc_src = pc[v];
  }
  __device__ __forceinline__ void edge(int owner, eoff_t k, bool valid) {
    const int32_t ps = big ? pc_src : __shfl(pc_src, owner, 64);
    if (valid) {
      const vid_t dst = __builtin_nontemporal_load(colidx + k);
      if (depth[dst] == next_level) atomicAdd(&pc[dst], ps);
    }
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
bc_push_kernel(const eoff_t *__restrict__ rowptr, int32_t m, int32_t level, ExpBigList big, BcPushVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vis.pc_src = 0;
  vis.big = 0;
  if (v < (unsigned)m && vis.depth[v] == level) {
    b = rowptr[v];
    e = rowptr[v + 1];
    vis.pc_src = vis.pc[v];
  }
  gdn_expand_wave(b, e, (vid_t)v, big, vis, s_scan[threadIdx.x >> 6]);
}

__global__ void __launch_bounds__(GDN_BLOCK)
bc_push_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, BcPushVis vis) {
  vis.big = 1;
  vis.pc_src = 0;
  gdn_expand_big_items(rowptr, big, vis);
}

// x of the forward sweep: the path count (raw bits) of the vertices at depth `level`, 0 elsewhere
__global__ void __launch_bounds__(GDN_BLOCK)
bc_x_fwd_kernel(const int32_t *__restrict__ depth, const int32_t *__restrict__ pc, int32_t m, int32_t level, float *__restrict__ x) {
  const size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (size_t)m) x[v] = depth[v] == level ? __int_as_float(pc[v]) : 0.0f;
}

// backward sweep: w[v] = (1 + delta[v]) / pc[v] of the vertices at depth next_level; first its maximum, then x = w * scale
__device__ __forceinline__ float bc_w(const bc_i32x4 r) { return __fdiv_rn(gdn_fadd(1.0f, __int_as_float(r.z)), (float)r.y); }

__global__ void __launch_bounds__(GDN_BLOCK)
bc_w_max_kernel(const bc_i32x4 *__restrict__ rec, int32_t m, int32_t next_level, unsigned *__restrict__ out) {
  float mx = 0.0f;
  for (size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; v < (size_t)m; v += (size_t)gridDim.x * GDN_BLOCK) {
    const bc_i32x4 r = rec[v];
    if (r.x == next_level) {
      const float w = bc_w(r);
      if (w > mx && w < 3.0e38f) mx = w;  // inf / nan (a wrapped path count of 0) do not take part in the sweep
    }
  }
  gdn_block_max_u32(mx > 0.0f ? __float_as_uint(mx) : 0u, out);
}

__global__ void __launch_bounds__(GDN_BLOCK)
bc_x_back_kernel(const bc_i32x4 *__restrict__ rec, int32_t m, int32_t next_level, float scale, float *__restrict__ x,
                 unsigned *__restrict__ odd) {
  const size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v >= (size_t)m) return;
  const bc_i32x4 r = rec[v];
  float xv = 0.0f;
  if (r.x == next_level) {
    const float w = bc_w(r);
    if (w >= 0.0f && w < 3.0e38f) xv = gdn_fmul(w, scale);
    else *odd = 1u;  // a non-finite term: this level falls back to the gather path (same arithmetic as the reference)
  }
  x[v] = xv;
}

// light backward level, vertex-parallel: the rows at depth `level` (see bc_back_kernel for the row classes)
__global__ void __launch_bounds__(GDN_BLOCK)
bc_back_all_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, int32_t m, int32_t level,
                   const int32_t *__restrict__ depth, bc_i32x4 *__restrict__ rec, float *__restrict__ scores,
                   vid_t *__restrict__ big_rows, BcCounters *cnt, unsigned cap) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  const int32_t next_level = level + 1;
  eoff_t b = 0, e = 0;
  float pcs = 0.0f;
  bool mine = false;
  if (i < (unsigned)m && depth[i] == level) {  // the 4-byte depth filters; only the level's vertices read their record
    mine = true;
    b = rowptr[i];
    e = rowptr[i + 1];
    pcs = (float)rec[i].y;
  }
  const vid_t v = (vid_t)i;
  const bool is_big = e - b >= BC_BLOCK_ROW;
  gdn_wl_push(big_rows, &cnt->big_count, cap, is_big, v, &cnt->overflow);
  const float acc = bc_back_wave_rows<false>(colidx, rec, next_level, b, e, pcs, is_big);
  if (mine && !is_big) {
    rec[v].z = __float_as_int(acc);
    scores[v] = gdn_fadd(scores[v], acc);
  }
}

struct gdn_bc_plan {
  const gdn_graph *g = nullptr;
  gdn_graph *gin_own = nullptr;  // transposed here when the caller has no in-CSR
  const gdn_graph *gin = nullptr;
  gdn_bfs_plan *bfs = nullptr;
  PbPlan fwd, back;              // layouts of the in-CSR / the out-CSR
  DevBuf<int32_t> depth, pc;
  DevBuf<bc_i32x4> rec;
  DevBuf<float> x;
  DevBuf<vid_t> big_rows;
  DevBuf<unsigned long long> bigitems;
  DevBuf<BcCounters> cnt;
  DevBuf<BcLevelStats> lstats;
  DevBuf<unsigned> mx;  // [0] max bits, [1] odd flag
  unsigned bigcap = 0, rowcap = 0;
  uint64_t max_deg = 1;
  double prep_ms = 0;
  ~gdn_bc_plan() {
    if (bfs) gdn_bfs_plan_free(bfs);
    if (gin_own) gdn_graph_free(gin_own);
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
bc_maxdeg_kernel(const eoff_t *__restrict__ rowptr, int32_t m, unsigned long long *__restrict__ out) {
  unsigned long long mx = 0;
  for (size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < (size_t)m; i += (size_t)gridDim.x * GDN_BLOCK) {
    const unsigned long long d = rowptr[i + 1] - rowptr[i];
    mx = d > mx ? d : mx;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(mx, o, 64);
    mx = t > mx ? t : mx;
  }
  if (gdn_lane() == 0 && mx) atomicMax(out, mx);
}

// no record tiers in BC's layouts; only the form of V travels (PbPlan::v_il: lane-interleaved blocks of 512 edges)
static inline PbMidArgs bc_mid_args(const PbPlan &pb) {
  PbMidArgs mid = PbMidArgs();
  mid.v_il = pb.v_il ? 1 : 0;
  return mid;
}

static int bc_pb_sweep_fwd(gdn_bc_plan &p, int32_t level) {
  PbPlan &pb = p.fwd;
  const int32_t m = p.g->m;
  hipLaunchKernelGGL(bc_x_fwd_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, p.depth.p, p.pc.p, m, level, p.x.p);
  const size_t lds_a = sizeof(float) * (pb.chunk_slots + 4);
  const size_t lds_b = sizeof(unsigned long long) << pb.log_bin;
  hipLaunchKernelGGL(pb_expand_kernel<0>, dim3(pb.nchunks), dim3(PB_THREADS), lds_a, 0, p.x.p, pb.m_global, pb.log_chunk, pb.chunk_ptr.p,
                     pb.chunk_order.p, pb.U.p, pb.G.p, pb.vals.p, pb.src_bits.p, pb.chunk_lo.p, 1u, pb.log_group, /*raw*/ 8,
                     nullptr, nullptr, nullptr, 0u, nullptr, pb.errflag.p, pb.chunk_slots);
  BcPcOp op;
  op.depth = p.depth.p;
  op.pc = p.pc.p;
  op.next_level = level + 1;
  op.vec_ok = false;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(pb_accumulate_kernel<BcPcOp>), dim3(pb.nbins), dim3(PB_THREADS), lds_b, 0, pb.m_local,
                     pb.log_bin, pb.bin_ptr.p, pb.bin_order.p, pb.V.p, pb.vals.p, pb.partial.p, pb.errflag.p, pb.dst_bits.p,
                     pb.bin_lo.p, op, 0, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                     bc_mid_args(pb));
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

// returns GDN_OK and *done = false when the level holds a non-finite term (the caller takes the gather path)
static int bc_pb_sweep_back(gdn_bc_plan &p, int32_t level, float *d_scores, bool *done) {
  PbPlan &pb = p.back;
  const int32_t m = p.g->m;
  *done = false;
  GDN_HIP(hipMemsetAsync(p.mx.p, 0, 8, 0));
  hipLaunchKernelGGL(bc_w_max_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, p.rec.p, m, level + 1, p.mx.p);
  unsigned h[2];
  GDN_HIP(hipMemcpy(h, p.mx.p, 8, hipMemcpyDeviceToHost));
  float wmax;
  memcpy(&wmax, &h[0], 4);
  if (!(wmax > 0.0f)) {  // no successor carries anything: every delta of the level is 0
    *done = true;
    return GDN_OK;
  }
  // scale = 2^-k with wmax * max_deg * 2^-k < 1: a row sum stays inside the unsigned 2^-62 fixed point
  int e1 = 0, e2 = 0;
  (void)frexpf(wmax, &e1);                 // wmax < 2^e1
  (void)frexp((double)p.max_deg, &e2);     // max_deg < 2^e2
  const int k = e1 + e2;
  const float scale = ldexpf(1.0f, -k), unscale = ldexpf(1.0f, k);
  hipLaunchKernelGGL(bc_x_back_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, p.rec.p, m, level + 1, scale, p.x.p,
                     p.mx.p + 1);
  GDN_HIP(hipMemcpy(h, p.mx.p, 8, hipMemcpyDeviceToHost));
  if (h[1]) return GDN_OK;  // inf / nan among the terms: not for the fixed-point sweep
  const size_t lds_a = sizeof(float) * (pb.chunk_slots + 4);
  const size_t lds_b = sizeof(unsigned long long) << pb.log_bin;
  hipLaunchKernelGGL(pb_expand_kernel<0>, dim3(pb.nchunks), dim3(PB_THREADS), lds_a, 0, p.x.p, pb.m_global, pb.log_chunk, pb.chunk_ptr.p,
                     pb.chunk_order.p, pb.U.p, pb.G.p, pb.vals.p, pb.src_bits.p, pb.chunk_lo.p, 1u, pb.log_group, 0, nullptr,
                     nullptr, nullptr, 0u, nullptr, pb.errflag.p, pb.chunk_slots);
  BcBackOp op;
  op.rec = p.rec.p;
  op.scores = d_scores;
  op.level = level;
  op.unscale = unscale;
  op.vec_ok = false;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(pb_accumulate_kernel<BcBackOp>), dim3(pb.nbins), dim3(PB_THREADS), lds_b, 0, pb.m_local,
                     pb.log_bin, pb.bin_ptr.p, pb.bin_order.p, pb.V.p, pb.vals.p, pb.partial.p, pb.errflag.p, pb.dst_bits.p,
                     pb.bin_lo.p, op, 0, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                     bc_mid_args(pb));
  GDN_HIP(hipGetLastError());
  *done = true;
  return GDN_OK;
}

extern "C" {

int gdn_bc_plan_create(const gdn_graph *g, const gdn_graph *gin, gdn_bc_plan **plan) {
  GDN_REQUIRE(plan != nullptr, "plan");
  *plan = nullptr;
  GDN_REQUIRE(g != nullptr, "graph");
  GDN_REQUIRE(gin == nullptr || gin->m == g->m, "in-CSR vertex count");
  HostTimer t;
  t.start();
  gdn_bc_plan *p = new gdn_bc_plan();
  p->g = g;
  int st = GDN_OK;
  do {
    if (!gin) {
      if ((st = gdn_graph_transpose(g, &p->gin_own))) break;
      gin = p->gin_own;
    }
    p->gin = gin;
    const int32_t m = g->m;
    if ((st = gdn_bfs_plan_create(g, gin, 1, &p->bfs))) break;
    int lc = 10, lb = 10;
    while (lc < PB_MAX_LOG_CHUNK && ((int64_t)1 << (lc + 9)) < (int64_t)m) lc++;  // slice sizes as for PageRank (pb_pick_log)
    while (lb < PB_MAX_LOG_BIN && ((int64_t)1 << (lb + 9)) < (int64_t)m) lb++;
    if (const char *e = gdn_xoption("GDN_BC_LOG_CHUNK")) lc = atoi(e) >= 10 && atoi(e) <= PB_MAX_LOG_CHUNK ? atoi(e) : lc;  // tuning knobs
    if (const char *e = gdn_xoption("GDN_BC_LOG_BIN")) lb = atoi(e) >= 10 && atoi(e) <= PB_MAX_LOG_BIN ? atoi(e) : lb;
    // forward: rows = destinations, columns = sources (the in-CSR); backward: rows = sources (the out-CSR)
    PbScratch scratch;
    // both layouts from the tiered builder's gather pass + LDS-staged splits (gdn_pbtier.hpp; no record tiers here: BC's
    // values are not fixed-point codes of one table); outside its limits, or GDN_PB_BUILDER=old: pb_build's key sort
    const char *vie = gdn_test_option("GDN_PB_V_IL"), *be = gdn_option("GDN_PB_BUILDER");
    const bool v_il = !(vie && vie[0] == '0');
    auto build_one = [&](const gdn_graph *src, PbPlan &out) -> int {
      if (!(be && be[0] == 'o') && lb <= PB_MID_ROW_BITS) {
        PbTieredArgs ta;
        PbTierSet ts;
        ta.rowptr = src->rowptr;
        ta.colidx = src->colidx;
        ta.m_raw = m;
        ta.m_rows = m;
        ta.m_global = m;
        ta.nnz = src->nnz;
        ta.src_count = nullptr;
        ta.log_chunk = lc;
        ta.log_bin = lb;
        ta.pad = 32;
        ta.log_group = 5;
        ta.tiers = false;
        ta.v_interleave = v_il;
        const int rc = pb_build_tiered_run(ta, out, ts);
        if (rc <= 0) return rc;  // built, or an error
      }
      const int rc = pb_build(src, m, lc, lb, out, true, nullptr, nullptr, /*compact=*/true, false, /*pad=*/32, /*log_group=*/5, nullptr,
                              0, false, false, nullptr, 0, false, false, PB_MAX_LOG_BIN, &scratch);
      if (rc != GDN_OK) return rc;
      return v_il ? pb_v_interleave(out) : GDN_OK;  // V in lane-interleaved blocks where the bins allow it (PbPlan::v_il)
    };
    // forward: rows = destinations, columns = sources (the in-CSR); backward: rows = sources (the out-CSR)
    if ((st = build_one(gin, p->fwd)) || (st = build_one(g, p->back))) break;
    const uint64_t bigcap64 = g->nnz / EXP_CHUNK + (uint64_t)m / 64 + 1024;
    p->bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
    p->rowcap = (unsigned)(g->nnz / BC_BLOCK_ROW + 16);
    if ((st = p->depth.alloc((size_t)m)) || (st = p->pc.alloc((size_t)m)) || (st = p->rec.alloc((size_t)m)) ||
        (st = p->x.alloc((size_t)m + 4)) || (st = p->big_rows.alloc(p->rowcap)) || (st = p->bigitems.alloc(p->bigcap)) ||
        (st = p->cnt.alloc(1)) || (st = p->lstats.alloc(1)) || (st = p->mx.alloc(2)))
      break;
    DevBuf<unsigned long long> md;
    if ((st = md.alloc(1))) break;
    (void)hipMemset(md.p, 0, 8);
    hipLaunchKernelGGL(bc_maxdeg_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, md.p);
    unsigned long long h_md = 1;
    if (hipMemcpy(&h_md, md.p, 8, hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_bc_plan_create: max degree readback failed");
      st = GDN_ERR_HIP;
      break;
    }
    p->max_deg = h_md ? h_md : 1;
    const int lds_a = (int)(sizeof(float) * ((p->fwd.chunk_slots > p->back.chunk_slots ? p->fwd.chunk_slots : p->back.chunk_slots) + 4));
    const int lds_b = (int)(sizeof(unsigned long long) << lb);
    hipError_t e = hipFuncSetAttribute((const void *)pb_expand_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)pb_accumulate_kernel<BcPcOp>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)pb_accumulate_kernel<BcBackOp>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
    if (e != hipSuccess) {
      gdn_set_error("gdn_bc_plan_create: hipFuncSetAttribute(dynamic LDS): %s", hipGetErrorString(e));
      st = GDN_ERR_HIP;
      break;
    }
    if (hipDeviceSynchronize() != hipSuccess) {
      gdn_set_error("gdn_bc_plan_create: %s", hipGetErrorString(hipGetLastError()));
      st = GDN_ERR_HIP;
    }
  } while (0);
  if (st != GDN_OK) {
    delete p;
    return st;
  }
  p->prep_ms = t.stop_ms();
  *plan = p;
  return GDN_OK;
}

int gdn_bc_plan_free(gdn_bc_plan *plan) {
  delete plan;
  return GDN_OK;
}

int gdn_bc_run(gdn_bc_plan *plan, int32_t source, float *d_scores, gdn_stats *stats) {
  GDN_REQUIRE(plan != nullptr && d_scores != nullptr, "plan / d_scores");
  gdn_bc_plan &p = *plan;
  const gdn_graph *g = p.g;
  const int32_t m = g->m;
  GDN_REQUIRE(source >= 0 && source < m, "source out of range");
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  st.prep_ms = p.prep_ms;
  HostTimer tsolve;
  tsolve.start();
  // depths: the BFS plan (unreached = MYINFINITY = 1e9, which no level equals)
  gdn_stats bst;
  GDN_TRY(gdn_bfs_run(p.bfs, source, p.depth.p, &bst));
  GDN_HIP(hipMemsetAsync(p.lstats.p, 0, sizeof(BcLevelStats), 0));
  hipLaunchKernelGGL(bc_level_stats_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, p.depth.p, g->rowptr, m, (int32_t)1000000000, p.lstats.p);
  BcLevelStats ls;
  GDN_HIP(hipMemcpy(&ls, p.lstats.p, sizeof(ls), hipMemcpyDeviceToHost));
  if (ls.deep) {  // a deep, thin graph: nothing is heavy there -- the queue-based path
    return gdn_bc_dev(g, source, d_scores, stats);
  }
  int32_t nlev = 0;
  for (int d = 0; d < BC_MAX_LEVELS; d++)
    if (ls.count[d]) nlev = d + 1;
  uint64_t heavy_div = 16;  // a level is heavy (two blocked sweeps instead of atomics / gathers) from nnz / heavy_div out-edges on
  if (const char *e = gdn_xoption("GDN_BC_HEAVY_DIV")) heavy_div = atoi(e) > 0 ? (uint64_t)atoi(e) : heavy_div;  // tuning knob
  const uint64_t heavy = g->nnz / heavy_div + 1;
  GDN_HIP(hipMemsetAsync(p.pc.p, 0, (size_t)m * 4, 0));
  GDN_HIP(hipMemsetAsync(p.cnt.p, 0, sizeof(BcCounters), 0));
  GDN_HIP(hipMemsetAsync(p.mx.p, 0, 8, 0));
  {
    const int32_t one = 1;
    GDN_HIP(hipMemcpyAsync(p.pc.p + source, &one, 4, hipMemcpyHostToDevice, 0));
  }
  ExpBigList big;
  big.items = p.bigitems.p;
  big.capacity = p.bigcap;
  big.count = &p.cnt.p->big_count;
  big.overflow = &p.cnt.p->overflow;
  // ---- forward: path counts level by level
  for (int32_t d = 0; d + 1 < nlev; d++) {
    if (ls.edges[d] >= heavy) {
      GDN_TRY(bc_pb_sweep_fwd(p, d));
    } else {
      BcPushVis vis;
      vis.colidx = g->colidx;
      vis.depth = p.depth.p;
      vis.pc = p.pc.p;
      vis.next_level = d + 1;
      vis.pc_src = 0;
      vis.big = 0;
      hipLaunchKernelGGL(bc_push_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, d, big, vis);
      hipLaunchKernelGGL(bc_push_big_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
      GDN_HIP(hipMemsetAsync(&p.cnt.p->big_count, 0, sizeof(unsigned), 0));
    }
  }
  // ---- backward
  hipLaunchKernelGGL(bc_pack_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, p.depth.p, p.pc.p, m, p.rec.p);
  for (int32_t d = nlev - 2; d >= 0; d--) {
    bool done = false;
    if (ls.edges[d] >= heavy) GDN_TRY(bc_pb_sweep_back(p, d, d_scores, &done));
    if (!done) {
      hipLaunchKernelGGL(bc_back_all_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, g->colidx, m, d,
                         p.depth.p, p.rec.p, d_scores, p.big_rows.p, p.cnt.p, p.rowcap);
      hipLaunchKernelGGL(bc_back_big_kernel, dim3(512), dim3(BC_BIG_THREADS), 0, 0, g->rowptr, g->colidx, p.big_rows.p, p.cnt.p,
                         p.rec.p, d_scores, d + 1, p.rowcap);
      GDN_HIP(hipMemsetAsync(&p.cnt.p->big_count, 0, sizeof(unsigned), 0));
    }
  }
  GDN_HIP(hipMemsetAsync(p.mx.p, 0, 4, 0));
  hipLaunchKernelGGL(bc_max_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, d_scores, m, p.mx.p);
  hipLaunchKernelGGL(bc_normalize_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, d_scores, m, p.mx.p);
  GDN_HIP(hipGetLastError());
  BcCounters h;
  GDN_HIP(hipMemcpy(&h, p.cnt.p, sizeof(h), hipMemcpyDeviceToHost));
  unsigned ef[2] = {0, 0};
  GDN_HIP(hipMemcpy(&ef[0], p.fwd.errflag.p, 4, hipMemcpyDeviceToHost));
  GDN_HIP(hipMemcpy(&ef[1], p.back.errflag.p, 4, hipMemcpyDeviceToHost));
  if (h.overflow || ef[0] || ef[1]) {
    gdn_set_error("gdn_bc_run: %s", h.overflow ? "device worklist overflow" : "a term left the fixed-point range of the blocked sweep");
    return GDN_ERR_OVERFLOW;
  }
  st.solve_ms = tsolve.stop_ms();
  st.iterations = nlev;
  uint64_t te = 0;
  for (int d = 0; d < nlev; d++) te += ls.edges[d];
  st.edges_traversed = 2 * te;
  if (stats) *stats = st;
  return GDN_OK;
}

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
  // the longest row decides which per-level launches exist at all: without a row of EXP_BIG edges no level has big-row
  // work items, without one of BC_BLOCK_ROW no backward level has rows for the workgroup kernel -- a road-like graph then
  // runs one launch per level and phase instead of two and a counter reset
  DevBuf<unsigned long long> d_maxdeg;
  GDN_TRY(d_maxdeg.alloc(1));
  GDN_HIP(hipMemsetAsync(d_maxdeg.p, 0, 8, 0));
  hipLaunchKernelGGL(bc_maxdeg_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, d_maxdeg.p);
  unsigned long long max_deg = 0;
  GDN_HIP(hipMemcpy(&max_deg, d_maxdeg.p, 8, hipMemcpyDeviceToHost));
  const bool fwd_big = max_deg >= EXP_BIG, back_big = max_deg >= BC_BLOCK_ROW;
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
  // frontiers of at most small_nf vertices and small_scout out-edges run fused in one workgroup (0 = never)
  unsigned small_nf = 1024;
  unsigned long long small_scout = 8192;
  if (const char *e = gdn_test_option("GDN_BC_SMALL_NF")) small_nf = (unsigned)atoi(e);  // tuning / test knobs
  if (const char *e = gdn_test_option("GDN_BC_SMALL_SCOUT")) small_scout = strtoull(e, nullptr, 10);
  DevBuf<BcSmallOut> small_out;
  if (small_nf) GDN_TRY(small_out.alloc(1));
  // levels of up to batch_nf vertices are queued fwd_batch at a time (1 = every level read back, the form for heavy levels)
  int fwd_batch = 8;
  unsigned batch_nf = 65536;
  if (const char *e = gdn_xoption("GDN_BC_FWD_BATCH")) fwd_batch = atoi(e) > 0 ? atoi(e) : 1;  // tuning / test knobs
  if (const char *e = gdn_xoption("GDN_BC_BATCH_NF")) batch_nf = (unsigned)atoi(e);
  DevBuf<unsigned> d_tails;
  GDN_TRY(d_tails.alloc((size_t)fwd_batch + 2));
  int mid_streak = 0;
  for (int32_t level = 0;;) {
    const unsigned l0 = lp[(size_t)level], nf = lp[(size_t)level + 1] - l0;
    if (nf == 0) break;
    if (small_nf && nf <= small_nf) {
      hipLaunchKernelGGL(bc_fwd_small_kernel, dim3(1), dim3(BC_SMALL_THREADS), 0, 0, g->rowptr, g->colidx, depth.p, pc.p, order.p,
                         l0, nf, (unsigned)m, level, small_nf, small_scout, cnt.p, small_out.p);
      unsigned hdr[2] = {0, 0};
      GDN_HIP(hipMemcpy(hdr, small_out.p, sizeof(hdr), hipMemcpyDeviceToHost));
      if (hdr[1]) {
        gdn_set_error("gdn_bc: device worklist overflow");
        return GDN_ERR_OVERFLOW;
      }
      if (hdr[0]) {
        const size_t at = lp.size();
        lp.resize(at + hdr[0]);
        GDN_HIP(hipMemcpy(lp.data() + at, small_out.p->tails, (size_t)hdr[0] * sizeof(unsigned), hipMemcpyDeviceToHost));
        level += (int32_t)hdr[0];
        continue;
      }
    }
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
    // (only behind a run of such levels: an R-MAT search passes through one mid-size level on its way to millions)
    mid_streak = nf <= batch_nf ? mid_streak + 1 : 0;
    if (fwd_batch > 1 && nf <= batch_nf && mid_streak > 4) {
      // ---- a batch of levels without a read-back in between (bc_fwd_lvl_kernel)
      std::vector<unsigned> hc((size_t)fwd_batch + 2, 0u);  // [0] start, [1] size of the first level, the rest zero
      hc[0] = l0;
      hc[1] = nf;
      GDN_HIP(hipMemcpyAsync(d_tails.p, hc.data(), hc.size() * sizeof(unsigned), hipMemcpyHostToDevice, 0));
      unsigned blocks = gdn_nblocks((uint64_t)nf * 4u);
      blocks = blocks < 64u ? 64u : (blocks > 2048u ? 2048u : blocks);
      for (int j = 0; j < fwd_batch; j++) {
        vis.next_level = level + 1 + j;
        hipLaunchKernelGGL(bc_fwd_lvl_kernel, dim3(blocks), dim3(GDN_BLOCK), 0, 0, g->rowptr, order.p, d_tails.p, j, big, vis);
        if (fwd_big) {
          hipLaunchKernelGGL(bc_fwd_lvl_big_kernel, dim3(256), dim3(GDN_BLOCK), 0, 0, g->rowptr, d_tails.p, j, big, vis);
          hipLaunchKernelGGL(bc_fwd_lvl_end_kernel, dim3(1), dim3(64), 0, 0, cnt.p);
        }
      }
      GDN_HIP(hipMemcpy(hc.data(), d_tails.p, hc.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
      GDN_HIP(hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost));
      if (h.overflow) {
        gdn_set_error("gdn_bc: device worklist overflow");
        return GDN_ERR_OVERFLOW;
      }
      std::vector<unsigned> ht((size_t)fwd_batch + 2);  // ht[j + 1] = end of level j of the batch
      ht[0] = l0;
      for (int j = 0; j <= fwd_batch; j++) ht[(size_t)j + 1] = ht[(size_t)j] + hc[(size_t)j + 1];
      {  // the per-level kernels and the fused one append at cnt->tail
        const unsigned t = ht[(size_t)fwd_batch + 1];
        GDN_HIP(hipMemcpyAsync(&cnt.p->tail, &t, sizeof(unsigned), hipMemcpyHostToDevice, 0));
        GDN_HIP(hipStreamSynchronize(0));
      }
      for (int j = 0; j < fwd_batch; j++) {  // the tails behind the levels that ran; an empty level ends the search
        lp.push_back(ht[(size_t)j + 2]);
        level++;
        if (ht[(size_t)j + 2] == ht[(size_t)j + 1]) break;
      }
      continue;
    }
    hipLaunchKernelGGL(bc_fwd_kernel, dim3(gdn_nblocks(nf)), dim3(GDN_BLOCK), 0, 0, g->rowptr, order.p + l0, nf, big, vis);
    if (fwd_big) hipLaunchKernelGGL(bc_fwd_big_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
    GDN_HIP(hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost));
    if (h.overflow) {
      gdn_set_error("gdn_bc: device worklist overflow");
      return GDN_ERR_OVERFLOW;
    }
    lp.push_back(h.tail);
    if (fwd_big) GDN_HIP(hipMemsetAsync(&cnt.p->big_count, 0, sizeof(unsigned), 0));
    level++;
  }
  const int32_t nlev = (int32_t)lp.size() - 2;  // non-empty levels 0 .. nlev-1
  // backward: the deepest level has no successors (its deltas stay 0, like the reference's first sweep)
  hipLaunchKernelGGL(bc_pack_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, depth.p, pc.p, m, rec.p);
  // runs of light levels (at most back_nf vertices, back_scout out-edges each) stay inside one workgroup
  unsigned back_nf = BC_SMALL_THREADS;
  unsigned long long back_scout = 8192;
  if (const char *e = gdn_test_option("GDN_BC_BACK_NF")) back_nf = std::min((unsigned)atoi(e), (unsigned)BC_SMALL_THREADS);  // tuning / test knobs
  if (const char *e = gdn_test_option("GDN_BC_BACK_SCOUT")) back_scout = strtoull(e, nullptr, 10);
  DevBuf<unsigned> d_lp;
  DevBuf<int32_t> d_next;
  if (back_nf && nlev >= 2) {
    GDN_TRY(d_lp.alloc(lp.size()));
    GDN_TRY(d_next.alloc(1));
    GDN_HIP(hipMemcpyAsync(d_lp.p, lp.data(), lp.size() * sizeof(unsigned), hipMemcpyHostToDevice, 0));
  }
  for (int32_t d = nlev - 2; d >= 0; d--) {
    const unsigned l0 = lp[(size_t)d], nf = lp[(size_t)d + 1] - l0;
    // (the fused kernel ends in a blocking read of where it stopped: worth it for a RUN of light levels -- the level
    // sizes are all known here -- not for one between two larger ones, which a single queued launch serves in ~5 us)
    int run = 0;
    if (back_nf && nf <= back_nf)
      for (int32_t dd = d; dd >= 0 && run < 8 && lp[(size_t)dd + 1] - lp[(size_t)dd] <= back_nf; dd--) run++;
    if (run >= 8 || (run > 0 && run == d + 1)) {
      hipLaunchKernelGGL(bc_back_small_kernel, dim3(1), dim3(BC_SMALL_THREADS), 0, 0, g->rowptr, g->colidx, order.p, d_lp.p, d, back_nf,
                         back_scout, rec.p, d_scores, d_next.p);
      int32_t next = d;
      GDN_HIP(hipMemcpy(&next, d_next.p, sizeof(next), hipMemcpyDeviceToHost));
      if (next < d) {  // levels d .. next + 1 are done
        d = next + 1;
        continue;
      }
    }
    hipLaunchKernelGGL(bc_back_kernel, dim3(gdn_nblocks(nf)), dim3(GDN_BLOCK), 0, 0, g->rowptr, g->colidx, order.p + l0, nf,
                       rec.p, d_scores, d + 1, big_rows.p, cnt.p, rowcap);
    if (back_big) {
      hipLaunchKernelGGL(bc_back_big_kernel, dim3(512), dim3(BC_BIG_THREADS), 0, 0, g->rowptr, g->colidx, big_rows.p, cnt.p, rec.p,
                         d_scores, d + 1, rowcap);
      GDN_HIP(hipMemsetAsync(&cnt.p->big_count, 0, sizeof(unsigned), 0));
    }
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
