// gdn_cc.hip -- connected components: Afforest (when the reverse graph is given) or
// Shiloach-Vishkin hooking + pointer jumping.
//
// Afforest (src/cc/omp_afforest.cc:37-83, CUDA src/cc/afforest.cu:32-88): two neighbour-sampling
// rounds link every vertex with its r-th neighbour (2m links instead of nnz), a sample of 1024 labels
// finds the giant component c, and only vertices OUTSIDE c walk their remaining out- and in-edges.
// On R-MAT graphs that skips almost every edge.  link() is a CAS loop that always points the higher
// root at the lower one, so the final label of a vertex is the minimum vertex id of its component --
// the same labels as the SV fixpoint -- whatever the sample is.  Label reads inside link() are
// agent-scope atomic loads: the 8 XCD L2s are not coherent for plain loads within one launch.
//
// Reference path: CCSolver (src/cc/cc.h:28).  OpenMP src/cc/omp_base.cc:6-50 (hook :24-37,
// shortcut :38-43, repeat while changed); CUDA src/cc/base.cu:8 hook (thread per vertex, racy
// plain store `comp[high] = low` :23-26), :31 shortcut, warp.cu:9 (warp per vertex).  Here the
// hook pass walks every edge through the load-balanced expansion of gdn_expand.hpp and links
// with a device-scope atomicMin, so labels only ever decrease and the fixpoint label of every
// vertex is the minimum vertex id of its component -- the same labels the reference produces
// (SURVEY 7 "Determinism"), independent of scheduling.  On a directed graph the hook is
// symmetric in (u,v) like omp_base.cc:27-36, so the out-CSR alone yields weakly connected
// components; in_csr is accepted for API parity and unused by this variant.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "gdn_expand.hpp"

struct CcCounters {
  unsigned changed;
  unsigned big_count;
  unsigned overflow;
  unsigned pad;
};

struct CcHookVis {
  const vid_t *__restrict__ colidx;
  int32_t *__restrict__ comp;
  CcCounters *cnt;
  int32_t cu;  // per-lane: label of this lane's vertex
  bool any;
  __device__ __forceinline__ void begin_big(vid_t v) { cu = comp[v]; }
  __device__ __forceinline__ void edge(int owner, eoff_t k, bool valid) {
    const int32_t cs = __shfl(cu, owner, 64);
    if (valid) {
      const vid_t dst = __builtin_nontemporal_load(colidx + k);
      const int32_t cd = comp[dst];
      if (cs != cd) {
        const int32_t high = cs > cd ? cs : cd;
        const int32_t low = cs + (cd - high);
        if (comp[high] == high) {  // omp_base.cc:33
          atomicMin(&comp[high], low);
          any = true;
        }
      }
    }
  }
  __device__ __forceinline__ void finish() {
    if (__ballot(any) && gdn_lane() == 0) cnt->changed = 1u;
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
cc_hook_kernel(const eoff_t *__restrict__ rowptr, int32_t m, ExpBigList big, CcHookVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vis.cu = 0;
  vis.any = false;
  if (v < (unsigned)m) {
    b = rowptr[v];
    e = rowptr[v + 1];
    vis.cu = vis.comp[v];
  }
  gdn_expand_wave(b, e, (vid_t)v, big, vis, s_scan[threadIdx.x >> 6]);
  vis.finish();
}

__global__ void __launch_bounds__(GDN_BLOCK)
cc_hook_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, CcHookVis vis) {
  vis.cu = 0;
  vis.any = false;
  gdn_expand_big_items(rowptr, big, vis);
  vis.finish();
}

// pointer jumping, omp_base.cc:38-43 / base.cu:31-37
__global__ void __launch_bounds__(GDN_BLOCK) cc_shortcut_kernel(int32_t *__restrict__ comp, int32_t m) {
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v >= (unsigned)m) return;
  int32_t c = comp[v];
  int32_t cc = comp[c];
  if (c == cc) return;
  while (c != cc) {
    c = cc;
    cc = comp[c];
  }
  comp[v] = c;
}

__device__ __forceinline__ int32_t cc_ld(const int32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------
// The "fusion" variant (src/cc/fusion.cu:47 cc_kernel: the whole Shiloach-Vishkin solve in ONE persistent kernel behind a
// software grid barrier, include/gbar.h): hook over every edge, barrier, pointer jumping, barrier, until a round changes
// nothing.  Cooperative grid, the ticket barrier of gdn_common.hpp; every label access is device-scope (the eight XCD L2s
// are not coherent within a launch).  A wavefront takes 64 consecutive rows and walks their edges 64 at a time, row by row.
// Kept for completeness of the reference's fusion row (GDN_CC_SV=fused): Afforest needs ONE sweep over the edges where
// this needs ~5, so it is not the default (RMAT-24: DESIGN 4.6).
// ctl: [0..2] the rotating "changed" flags of rounds i % 3, [3] rounds run.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GDN_BLOCK)
cc_sv_fused_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, int32_t m, int32_t *comp, unsigned *ctl,
                   unsigned *bar) {
  const unsigned lane = gdn_lane();
  const uint64_t wave = ((uint64_t)blockIdx.x * GDN_BLOCK + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * GDN_BLOCK) >> 6;
  const uint64_t tid = (uint64_t)blockIdx.x * GDN_BLOCK + threadIdx.x, nthreads = (uint64_t)gridDim.x * GDN_BLOCK;
  for (unsigned round = 0;; round++) {
    bool any = false;
    for (uint64_t v0 = wave * 64; v0 < (uint64_t)m; v0 += nwaves * 64) {  // hook, src/cc/omp_base.cc:24-37
      const uint64_t v = v0 + lane;
      eoff_t b = 0, e = 0;
      int32_t cv = 0;
      if (v < (uint64_t)m) {
        b = rowptr[v];
        e = rowptr[v + 1];
        cv = cc_ld(comp + v);
      }
      unsigned long long rows = __ballot(e > b);
      while (rows) {
        const int l = __ffsll((long long)rows) - 1;
        rows &= rows - 1;
        const eoff_t rb = ((eoff_t)(unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)b, l);
        const eoff_t re = ((eoff_t)(unsigned)__builtin_amdgcn_readlane((int)(e >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)e, l);
        const int32_t cs = __builtin_amdgcn_readlane(cv, l);
        for (eoff_t k = rb + lane; k < re; k += 64) {
          const int32_t cd = cc_ld(comp + colidx[k]);
          if (cs != cd) {
            const int32_t high = cs > cd ? cs : cd, low = cs + (cd - high);
            if (cc_ld(comp + high) == high) {  // omp_base.cc:33
              atomicMin(comp + high, low);
              any = true;
            }
          }
        }
      }
    }
    if (__ballot(any) && lane == 0) __hip_atomic_store(ctl + round % 3u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gdn_grid_barrier(bar, gridDim.x);
    for (uint64_t v = tid; v < (uint64_t)m; v += nthreads) {  // shortcut, omp_base.cc:38-43
      int32_t c = cc_ld(comp + v), cc2 = cc_ld(comp + c);
      if (c != cc2) {
        while (c != cc2) {
          c = cc2;
          cc2 = cc_ld(comp + c);
        }
        __hip_atomic_store(comp + v, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (tid == 0) {
      __hip_atomic_store(ctl + (round + 1u) % 3u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the next round's flag
      __hip_atomic_store(ctl + 3, round + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gdn_grid_barrier(bar, gridDim.x);
    if (__hip_atomic_load(ctl + round % 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) break;  // (the same word for every workgroup)
  }
}

// ------------------------------------------------------------------------------------------
// Afforest
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ void cc_link(int32_t u, int32_t v, int32_t *comp) {  // omp_afforest.cc:12-25
  int32_t p1 = cc_ld(comp + u), p2 = cc_ld(comp + v);
  while (p1 != p2) {
    const int32_t high = p1 > p2 ? p1 : p2;
    const int32_t low = p1 + (p2 - high);
    const int32_t p_high = cc_ld(comp + high);
    if (p_high == low) break;
    if (p_high == high && atomicCAS(comp + high, high, low) == high) break;
    p1 = cc_ld(comp + cc_ld(comp + high));
    p2 = cc_ld(comp + low);
  }
}

// cc_link for every lane of a wave at once (has = this lane holds an edge).  Once trees are few and large (R-MAT after the
// first sampling round: stars around the lowest ids) the links that still join two of them are the SAME (root, root) pair in
// thousands of lanes at a time -- every one a compare-and-swap on one address, served one after the other (RMAT-24: sampling
// round 1 took 1.16 ms against round 0's 0.25).  One lane per distinct pair of a wave issues it; the others find the link made
// when they walk cc_link's loop afterwards.  Labels do not depend on who wins a link (the final labels are the minimum ids).
__device__ __forceinline__ void cc_link_wave(bool has, int32_t u, int32_t w, int32_t *comp) {
  int32_t p1 = 0, p2 = 0;
  if (has) {
    p1 = cc_ld(comp + u);
    p2 = cc_ld(comp + w);
  }
  const bool need = has && p1 != p2;
  const int32_t high = p1 > p2 ? p1 : p2, low = p1 + (p2 - high);
  const bool want = need && cc_ld(comp + high) == high;
  unsigned long long wm = __ballot(want);
  const unsigned lane = gdn_lane();
  while (wm) {
    const int leader = __ffsll((long long)wm) - 1;
    const int32_t h = __shfl(high, leader, 64), l = __shfl(low, leader, 64);
    const unsigned long long grp = __ballot(want && high == h && low == l);
    if (lane == (unsigned)leader) atomicCAS(comp + h, h, l);
    wm &= ~grp;
  }
  if (need) cc_link(u, w, comp);
}

// sampling round r: link v with its r-th out-neighbour (omp_afforest.cc:40-46)
__global__ void __launch_bounds__(GDN_BLOCK)
cc_sample_link_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, int32_t m, int r,
                      int32_t *__restrict__ comp, unsigned v0 = 0) {
  const unsigned v = v0 + blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  if (v < (unsigned)m) {
    b = rowptr[v];
    e = rowptr[v + 1];
  }
  const bool has = b + (eoff_t)r < e;
  const int32_t w = has ? colidx[b + r] : 0;
  if (r == 0) {
    if (has) cc_link((int32_t)v, w, comp);
    return;
  }
  cc_link_wave(has, (int32_t)v, w, comp);
}

__global__ void __launch_bounds__(GDN_BLOCK)
cc_sample_labels_kernel(const int32_t *__restrict__ comp, int32_t m, int32_t *__restrict__ out, int n) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i >= (unsigned)n) return;
  unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
  z ^= z >> 29;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 32;
  out[i] = comp[z % (unsigned long long)m];
}

// the most frequent label of the sample (ties: the smallest label, like a scan of the sorted sample keeping the first
// maximum) -> *out, on the device: the closing passes read it from there, no copy back and host sort in between (50 us of a
// 1 ms solve)
__global__ void __launch_bounds__(1024) cc_sample_mode_kernel(const int32_t *__restrict__ sample, int n, int32_t *__restrict__ out) {
  __shared__ int32_t s_lab[1024];
  __shared__ unsigned long long s_best[16];
  const int i = (int)threadIdx.x;
  s_lab[i] = i < n ? sample[i] : -1;
  __syncthreads();
  unsigned long long key = 0ull;
  if (i < n) {
    const int32_t mine = s_lab[i];
    unsigned cnt = 0;
    for (int j = 0; j < n; j++) cnt += s_lab[j] == mine ? 1u : 0u;
    key = ((unsigned long long)cnt << 32) | (unsigned)(0x7FFFFFFF - mine);  // more often first, then the smaller label
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(key, o, 64);
    key = t > key ? t : key;
  }
  if ((i & 63) == 0) s_best[i >> 6] = key;
  __syncthreads();
  if (i == 0) {
    for (int w = 1; w < 16; w++) key = s_best[w] > key ? s_best[w] : key;
    *out = 0x7FFFFFFF - (int32_t)(unsigned)(key & 0xFFFFFFFFull);
  }
}

struct CcLinkVis {
  const vid_t *__restrict__ colidx;
  int32_t *__restrict__ comp;
  int32_t v;  // per-lane source vertex
  __device__ __forceinline__ void begin_big(vid_t vv) { v = vv; }
  __device__ __forceinline__ void edge(int owner, eoff_t k, bool valid) {
    const int32_t src = __shfl(v, owner, 64);
    cc_link_wave(valid, src, valid ? colidx[k] : 0, comp);
  }
};

// remaining edges of the vertices outside the giant component c (omp_afforest.cc:56-76);
// skip = neighbours already used by the sampling rounds (out-CSR) or 0 (in-CSR)
__global__ void __launch_bounds__(GDN_BLOCK)
cc_finish_kernel(const eoff_t *__restrict__ rowptr, int32_t m, const int32_t *__restrict__ c_ptr, int skip, ExpBigList big,
                 CcLinkVis vis) {
  const int32_t c = c_ptr ? *c_ptr : -1;  // label of the giant component (nullptr: nobody is skipped)
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vis.v = (int32_t)v;
  if (v < (unsigned)m && vis.comp[v] != c) {
    b = rowptr[v];
    e = rowptr[v + 1];
    // a big row is cut into work items counted from the row's FIRST edge (gdn_expand_big_items): it keeps its sampled
    // neighbours (re-linking them is harmless) -- skipping them here would shorten the item count and lose the row's
    // last `skip` edges
    if (e - b < big.min_deg) b = (b + (eoff_t)skip < e) ? b + skip : e;
  }
  gdn_expand_wave(b, e, (vid_t)v, big, vis, s_scan[threadIdx.x >> 6]);
}

// big rows: the chunk work item does not know the skip; re-linking 2 sampled neighbours is harmless
__global__ void __launch_bounds__(GDN_BLOCK)
cc_finish_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, CcLinkVis vis) {
  vis.v = 0;
  gdn_expand_big_items(rowptr, big, vis);
}

// ---- out-edges only (no reverse graph): the skip of the giant component c made safe.  An edge whose two ends were in c's
// tree when the sampling rounds ended needs no link -- trees only ever merge -- and such edges are nearly all of an R-MAT
// graph; what must be linked are the edges with an end OUTSIDE c (the in-edges of a vertex outside c included, which is what
// Afforest reads the reverse graph for).  One bit per vertex says "outside c" (2 MB at RMAT-24: the test of an edge's target
// is a gather from the L2, not from a 64 MB label array), and the closing pass streams every edge but links few.
__global__ void __launch_bounds__(GDN_BLOCK)
cc_outside_bits_kernel(const int32_t *__restrict__ comp, int32_t m, const int32_t *__restrict__ c_ptr, unsigned long long *__restrict__ bits) {
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  const int32_t c = *c_ptr;
  const unsigned long long mask = __ballot(v < (unsigned)m && comp[v] != c);
  if (gdn_lane() == 0 && v < (unsigned)m) bits[v >> 6] = mask;
}
struct CcLinkOutsideVis {
  const vid_t *__restrict__ colidx;
  int32_t *__restrict__ comp;
  const unsigned long long *__restrict__ bits;
  int32_t v;   // per-lane source vertex
  int vout;    // per-lane: the source is outside c
  __device__ __forceinline__ void begin_big(vid_t vv) {
    v = vv;
    vout = (int)((bits[(unsigned)vv >> 6] >> ((unsigned)vv & 63u)) & 1ull);
  }
  __device__ __forceinline__ void edge(int owner, eoff_t k, bool valid) {
    const int32_t src = __shfl(v, owner, 64);
    const int so = __shfl(vout, owner, 64);
    const vid_t dst = valid ? colidx[k] : 0;
    const bool need = valid && (so || ((bits[(unsigned)dst >> 6] >> ((unsigned)dst & 63u)) & 1ull));
    cc_link_wave(need, src, dst, comp);
  }
};
__global__ void __launch_bounds__(GDN_BLOCK)
cc_finish_outside_kernel(const eoff_t *__restrict__ rowptr, int32_t m, int skip, ExpBigList big, CcLinkOutsideVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vis.v = (int32_t)v;
  vis.vout = 0;
  if (v < (unsigned)m) {
    b = rowptr[v];
    e = rowptr[v + 1];
    vis.vout = (int)((vis.bits[v >> 6] >> (v & 63u)) & 1ull);
    if (e - b < big.min_deg) b = (b + (eoff_t)skip < e) ? b + skip : e;  // (see cc_finish_kernel)
  }
  gdn_expand_wave(b, e, (vid_t)v, big, vis, s_scan[threadIdx.x >> 6]);
}
__global__ void __launch_bounds__(GDN_BLOCK)
cc_finish_outside_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, CcLinkOutsideVis vis) {
  vis.v = 0;
  vis.vout = 0;
  gdn_expand_big_items(rowptr, big, vis);
}

__global__ void __launch_bounds__(GDN_BLOCK) cc_init_kernel(int32_t *__restrict__ comp, int32_t m) {
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (unsigned)m) comp[v] = (int32_t)v;
}

extern "C" {

static int cc_afforest(const gdn_graph *g, const gdn_graph *gin, int32_t *d_comp, gdn_stats *stats);

int gdn_cc_dev(const gdn_graph *g, const gdn_graph *gin, int32_t *d_comp, gdn_stats *stats) {
  GDN_REQUIRE(g != nullptr && d_comp != nullptr, "graph / d_comp");
  GDN_REQUIRE(gin == nullptr || gin->m == g->m, "in-CSR vertex count");
  // With the reverse graph (or the graph itself for a symmetric one) every edge can be seen from both ends: Afforest
  // with its skip of the giant component.  Without it (a directed graph, out-edges only) the skip is not safe -- an edge
  // from inside the giant component to a vertex outside it would never be linked -- but everything else is: the two
  // sampling rounds, then ONE pass that links the remaining out-edges of EVERY vertex (a link whose two ends already
  // share a root costs two loads).  That is a single sweep over the edges instead of Shiloach-Vishkin's ~5 (RMAT-24:
  // 21.9 ms); GDN_CC_SV=1 keeps the SV rounds (the reference's src/cc/omp_base.cc algorithm) for comparison.
  const char *sv = gdn_option("GDN_CC_SV");
  if ((!sv || sv[0] == '0') && gin == nullptr) {
    // GDN_CC_REVERSE=build: the reverse graph built inside the call (gdn_graph_transpose: the round-4 partition kernels) and
    // charged to prep_ms, as the reference keeps its own graph preparation outside the Timer -- the solve then is the one
    // WITH the reverse graph (its skip of the giant component).  Off by default: wall time (prep + solve) is lower without.
    const char *rv = gdn_option("GDN_CC_REVERSE");
    if (rv && rv[0] == 'b' && g->nnz > 0) {
      HostTimer tbuild;
      tbuild.start();
      gdn_graph *rev = nullptr;
      GDN_TRY(gdn_graph_transpose(g, &rev));
      GDN_HIP(hipDeviceSynchronize());
      const double build_ms = tbuild.stop_ms();
      gdn_stats st;
      memset(&st, 0, sizeof(st));
      const int rc = cc_afforest(g, rev, d_comp, &st);
      gdn_graph_free(rev);
      st.prep_ms += build_ms;
      st.reserved = 3;  // (3 = the reverse graph was built inside the call)
      if (rc == GDN_OK && stats) *stats = st;
      return rc;
    }
  }
  if (!sv || sv[0] == '0') return cc_afforest(g, gin, d_comp, stats);
  if (gin != nullptr && sv[0] != 'f') return cc_afforest(g, gin, d_comp, stats);
  const int32_t m = g->m;
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  HostTimer tprep, tsolve;
  tprep.start();
  if (sv[0] == 'f') {  // GDN_CC_SV=fused: the whole solve in one cooperative launch (cc_sv_fused_kernel)
    int dev = 0, coop = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) == hipSuccess && coop &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cc_sv_fused_kernel, GDN_BLOCK, 0) == hipSuccess && per_cu >= 1) {
      DevBuf<unsigned> ctl, bar;
      GDN_TRY(ctl.alloc(4));
      GDN_TRY(bar.alloc(GDN_GBAR_WORDS));
      st.prep_ms = tprep.stop_ms();
      tsolve.start();
      hipLaunchKernelGGL(cc_init_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, d_comp, m);
      GDN_HIP(hipMemsetAsync(ctl.p, 0, 4 * sizeof(unsigned), 0));
      GDN_HIP(hipMemsetAsync(bar.p, 0, GDN_GBAR_WORDS * sizeof(unsigned), 0));
      const eoff_t *a_rowptr = g->rowptr;
      const vid_t *a_colidx = g->colidx;
      int32_t a_m = m;
      int32_t *a_comp = d_comp;
      unsigned *a_ctl = ctl.p, *a_bar = bar.p;
      void *args[] = {&a_rowptr, &a_colidx, &a_m, &a_comp, &a_ctl, &a_bar};
      // (one workgroup fewer per CU than the occupancy query admits, at most 4: MI355X_MICROARCH.md, residency)
      const int k = per_cu > 4 ? 4 : (per_cu > 1 ? per_cu - 1 : 1);
      GDN_HIP(hipLaunchCooperativeKernel((const void *)cc_sv_fused_kernel, dim3((unsigned)(cus * k)), dim3(GDN_BLOCK), args, 0, 0));
      unsigned h[4] = {0, 0, 0, 0};
      GDN_HIP(hipMemcpy(h, ctl.p, sizeof(h), hipMemcpyDeviceToHost));
      st.solve_ms = tsolve.stop_ms();
      st.iterations = (int32_t)h[3];
      st.edges_traversed = g->nnz * (uint64_t)h[3];
      st.reserved = 2;  // (2 = the fused Shiloach-Vishkin kernel ran)
      if (stats) *stats = st;
      return GDN_OK;
    }
    (void)hipGetLastError();  // no cooperative launch on this device: the rounds as launches, below
  }
  DevBuf<unsigned long long> bigitems;
  DevBuf<CcCounters> cnt;
  const uint64_t bigcap64 = g->nnz / EXP_CHUNK + (uint64_t)m / 64 + 1024;
  const unsigned bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
  GDN_TRY(bigitems.alloc(bigcap));
  GDN_TRY(cnt.alloc(1));
  st.prep_ms = tprep.stop_ms();

  tsolve.start();  // omp_base.cc:19 (comp[n] = n is inside the reference solver too, :15)
  hipLaunchKernelGGL(cc_init_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, d_comp, m);
  ExpBigList big;
  big.items = bigitems.p;
  big.capacity = bigcap;
  big.count = &cnt.p->big_count;
  big.overflow = &cnt.p->overflow;
  CcHookVis vis;
  vis.colidx = g->colidx;
  vis.comp = d_comp;
  vis.cnt = cnt.p;
  vis.cu = 0;
  vis.any = false;
  int iter = 0;
  CcCounters h;
  for (;;) {
    ++iter;
    GDN_HIP(hipMemsetAsync(cnt.p, 0, sizeof(CcCounters), 0));
    hipLaunchKernelGGL(cc_hook_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, big, vis);
    hipLaunchKernelGGL(cc_hook_big_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
    hipLaunchKernelGGL(cc_shortcut_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, d_comp, m);
    GDN_HIP(hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost));
    if (h.overflow) {
      gdn_set_error("gdn_cc: device worklist overflow");
      return GDN_ERR_OVERFLOW;
    }
    if (!h.changed) break;
  }
  GDN_HIP(hipGetLastError());
  st.solve_ms = tsolve.stop_ms();
  st.iterations = iter;
  st.edges_traversed = g->nnz * (uint64_t)iter;
  if (stats) *stats = st;
  return GDN_OK;
}

}  // extern "C"

static int cc_afforest(const gdn_graph *g, const gdn_graph *gin, int32_t *d_comp, gdn_stats *stats) {
  const int32_t m = g->m;
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  HostTimer tprep, tsolve;
  tprep.start();
  DevBuf<unsigned long long> bigitems;
  DevBuf<CcCounters> cnt;
  DevBuf<int32_t> d_sample;
  const int nsample = 1024;  // src/cc/verifier.cc:13 SampleFrequentElement(num_samples = 1024)
  const uint64_t nn = (gin && gin->nnz > g->nnz) ? gin->nnz : g->nnz;
  const uint64_t bigcap64 = nn / EXP_CHUNK + (uint64_t)m / 64 + 1024;
  const unsigned bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
  GDN_TRY(bigitems.alloc(bigcap));
  GDN_TRY(cnt.alloc(1));
  GDN_TRY(d_sample.alloc(nsample + 1));  // + the label of the giant component
  DevBuf<unsigned long long> outside;    // out-edges only: one bit per vertex, "outside the giant component"
  if (!gin) GDN_TRY(outside.alloc(((size_t)m + 63) / 64 + 1));
  st.prep_ms = tprep.stop_ms();

  tsolve.start();
  const dim3 grid_m(gdn_nblocks((uint64_t)m)), blk(GDN_BLOCK);
  hipLaunchKernelGGL(cc_init_kernel, grid_m, blk, 0, 0, d_comp, m);
  const int neighbor_rounds = 2;  // omp_afforest.cc:37
  for (int r = 0; r < neighbor_rounds; r++) {
    // rounds after the first in two launches: a small head of the vertex range joins the few large trees round 0 left
    // (their roots are what thousands of waves would otherwise compare-and-swap at the same time, see cc_link_wave); the
    // rest then finds them joined.  RMAT-24, the whole solve: 1.48 -> 1.02 ms with a head of 2^10 .. 2^14 vertices (2^18: 1.19, 2^20: 1.48)
    unsigned head = 0;
    if (r > 0 && (unsigned)m > (1u << 18)) {
      head = 1u << 14;
      if (const char *e = gdn_xoption("GDN_CC_HEAD")) head = (unsigned)atoi(e) & ~(unsigned)(GDN_BLOCK - 1);  // tuning knob
      if (head >= (unsigned)m) head = 0;
    }
    if (head) {
      hipLaunchKernelGGL(cc_sample_link_kernel, dim3(head / GDN_BLOCK), blk, 0, 0, g->rowptr, g->colidx, (int32_t)head, r, d_comp, 0u);
      hipLaunchKernelGGL(cc_sample_link_kernel, dim3(gdn_nblocks((uint64_t)m - head)), blk, 0, 0, g->rowptr, g->colidx, m, r, d_comp, head);
    } else {
      hipLaunchKernelGGL(cc_sample_link_kernel, grid_m, blk, 0, 0, g->rowptr, g->colidx, m, r, d_comp, 0u);
    }
    hipLaunchKernelGGL(cc_shortcut_kernel, grid_m, blk, 0, 0, d_comp, m);
  }
  // most frequent label of the sample = the giant intermediate component
  hipLaunchKernelGGL(cc_sample_labels_kernel, dim3(gdn_nblocks(nsample)), blk, 0, 0, d_comp, m, d_sample.p, nsample);
  hipLaunchKernelGGL(cc_sample_mode_kernel, dim3(1), dim3(1024), 0, 0, d_sample.p, nsample, d_sample.p + nsample);
  GDN_HIP(hipMemsetAsync(cnt.p, 0, sizeof(CcCounters), 0));
  ExpBigList big;
  big.items = bigitems.p;
  big.capacity = bigcap;
  big.count = &cnt.p->big_count;
  big.overflow = &cnt.p->overflow;
  CcLinkVis vis;
  vis.comp = d_comp;
  vis.v = 0;
  vis.colidx = g->colidx;
  const char *oe = gdn_xoption("GDN_CC_OUTSIDE");  // 0: the unfiltered closing pass of round 3 (A/B)
  if (!gin && !(oe && oe[0] == '0')) {
    // out-edges only: every edge is streamed, the ones with an end outside c are linked (see cc_outside_bits_kernel)
    hipLaunchKernelGGL(cc_outside_bits_kernel, grid_m, blk, 0, 0, d_comp, m, d_sample.p + nsample, outside.p);
    CcLinkOutsideVis ov;
    ov.colidx = g->colidx;
    ov.comp = d_comp;
    ov.bits = outside.p;
    ov.v = 0;
    ov.vout = 0;
    hipLaunchKernelGGL(cc_finish_outside_kernel, grid_m, blk, 0, 0, g->rowptr, m, neighbor_rounds, big, ov);
    hipLaunchKernelGGL(cc_finish_outside_big_kernel, dim3(1024), blk, 0, 0, g->rowptr, big, ov);
  } else {
    // gin == nullptr (out-edges only): nobody is skipped (label -1 matches no vertex)
    hipLaunchKernelGGL(cc_finish_kernel, grid_m, blk, 0, 0, g->rowptr, m, gin ? d_sample.p + nsample : nullptr, neighbor_rounds, big, vis);
    hipLaunchKernelGGL(cc_finish_big_kernel, dim3(1024), blk, 0, 0, g->rowptr, big, vis);
  }
  if (gin && gin != g) {  // directed: the in-edges too (omp_afforest.cc:72-74)
    GDN_HIP(hipMemsetAsync(&cnt.p->big_count, 0, sizeof(unsigned), 0));
    vis.colidx = gin->colidx;
    hipLaunchKernelGGL(cc_finish_kernel, grid_m, blk, 0, 0, gin->rowptr, m, d_sample.p + nsample, 0, big, vis);
    hipLaunchKernelGGL(cc_finish_big_kernel, dim3(1024), blk, 0, 0, gin->rowptr, big, vis);
  }
  hipLaunchKernelGGL(cc_shortcut_kernel, grid_m, blk, 0, 0, d_comp, m);
  CcCounters h;
  GDN_HIP(hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost));
  if (h.overflow) {
    gdn_set_error("gdn_cc: device worklist overflow");
    return GDN_ERR_OVERFLOW;
  }
  GDN_HIP(hipGetLastError());
  st.solve_ms = tsolve.stop_ms();
  st.iterations = neighbor_rounds + 1;
  st.edges_traversed = g->nnz;
  if (stats) *stats = st;
  return GDN_OK;
}

extern "C" {

// Host API: one call == CCSolver(g, comp) (src/cc/main.cc:16).
int gdn_cc(int32_t m, uint64_t nnz, const uint64_t *rowptr, const int32_t *colidx, const uint64_t *in_rowptr,
           const int32_t *in_colidx, int32_t *comp, gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && rowptr && comp, "null argument");
  GDN_TRY(gdn_require_device());
  HostTimer th2d;
  th2d.start();
  gdn_graph *g = nullptr, *gi = nullptr;
  GDN_TRY(gdn_graph_upload(m, nnz, rowptr, colidx, &g));
  if (in_rowptr && in_colidx) {
    if (in_rowptr == rowptr && in_colidx == colidx) gi = g;  // symmetric graph (csr_graph.h:241-245)
    else {
      const int rc0 = gdn_graph_upload(m, nnz, in_rowptr, in_colidx, &gi);
      if (rc0 != GDN_OK) {
        gdn_graph_free(g);
        return rc0;
      }
    }
  }
  DevBuf<int32_t> d_comp;
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  int rc = GDN_OK;
  do {
    if ((rc = d_comp.alloc(m))) break;
    const double h2d = th2d.stop_ms();
    if ((rc = gdn_cc_dev(g, gi, d_comp.p, &st))) break;
    st.h2d_ms = h2d;
    if (hipMemcpy(comp, d_comp.p, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_cc: download failed");
      rc = GDN_ERR_HIP;
    }
  } while (0);
  if (gi && gi != g) gdn_graph_free(gi);
  gdn_graph_free(g);
  if (stats) *stats = st;
  return rc;
}

}  // extern "C"
