// gdn_cc.hip -- connected components: Shiloach-Vishkin hooking + pointer jumping.
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
#include <string.h>

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

__global__ void __launch_bounds__(GDN_BLOCK) cc_init_kernel(int32_t *__restrict__ comp, int32_t m) {
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (unsigned)m) comp[v] = (int32_t)v;
}

extern "C" {

int gdn_cc_dev(const gdn_graph *g, const gdn_graph *gin, int32_t *d_comp, gdn_stats *stats) {
  GDN_REQUIRE(g != nullptr && d_comp != nullptr, "graph / d_comp");
  (void)gin;
  const int32_t m = g->m;
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  HostTimer tprep, tsolve;
  tprep.start();
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

// Host API: one call == CCSolver(g, comp) (src/cc/main.cc:16).
int gdn_cc(int32_t m, uint64_t nnz, const uint64_t *rowptr, const int32_t *colidx, const uint64_t *in_rowptr,
           const int32_t *in_colidx, int32_t *comp, gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && rowptr && comp, "null argument");
  (void)in_rowptr;
  (void)in_colidx;
  GDN_TRY(gdn_require_device());
  HostTimer th2d;
  th2d.start();
  gdn_graph *g = nullptr;
  GDN_TRY(gdn_graph_upload(m, nnz, rowptr, colidx, &g));
  DevBuf<int32_t> d_comp;
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  int rc = GDN_OK;
  do {
    if ((rc = d_comp.alloc(m))) break;
    const double h2d = th2d.stop_ms();
    if ((rc = gdn_cc_dev(g, nullptr, d_comp.p, &st))) break;
    st.h2d_ms = h2d;
    if (hipMemcpy(comp, d_comp.p, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_cc: download failed");
      rc = GDN_ERR_HIP;
    }
  } while (0);
  gdn_graph_free(g);
  if (stats) *stats = st;
  return rc;
}

}  // extern "C"
