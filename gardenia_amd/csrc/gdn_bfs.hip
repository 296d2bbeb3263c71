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
#include <string.h>

#include "gdn_expand.hpp"

struct BfsCounters {  // device, zeroed per level by the host-side memset
  unsigned next_count;
  unsigned big_count;
  unsigned overflow;
  unsigned pad;
  unsigned long long scout;  // sum of out-degrees of the vertices discovered this level
  unsigned long long awake;  // vertices discovered by a bottom-up step
};

struct BfsTdVis {
  const eoff_t *__restrict__ rowptr;
  const vid_t *__restrict__ colidx;
  unsigned *__restrict__ visited;
  int32_t *__restrict__ depth;
  vid_t *__restrict__ outq;
  BfsCounters *cnt;
  unsigned cap;
  int32_t next_level;
  unsigned long long scout_local;
  __device__ __forceinline__ void begin_big(vid_t) {}
  __device__ __forceinline__ void edge(int, eoff_t k, bool valid) {
    bool claim = false;
    vid_t dst = 0;
    if (valid) {
      dst = __builtin_nontemporal_load(colidx + k);
      const unsigned bit = 1u << (dst & 31);
      const unsigned w = visited[dst >> 5];
      if (!(w & bit)) {
        const unsigned old = atomicOr(&visited[dst >> 5], bit);
        claim = !(old & bit);
      }
    }
    if (claim) {
      depth[dst] = next_level;
      scout_local += rowptr[dst + 1] - rowptr[dst];
    }
    gdn_wl_push(outq, &cnt->next_count, cap, claim, dst, &cnt->overflow);
  }
  __device__ __forceinline__ void finish() {
    const unsigned long long s = gdn_wave_sum(scout_local);
    if (gdn_lane() == 0 && s) atomicAdd(&cnt->scout, s);
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
bfs_td_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ inq, unsigned nf, ExpBigList big,
              BfsTdVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vid_t v = 0;
  if (i < nf) {
    v = inq[i];
    b = rowptr[v];
    e = rowptr[v + 1];
  }
  vis.scout_local = 0;
  gdn_expand_wave(b, e, v, big, vis, s_scan[threadIdx.x >> 6]);
  vis.finish();
}

__global__ void __launch_bounds__(GDN_BLOCK)
bfs_td_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, BfsTdVis vis) {
  vis.scout_local = 0;
  gdn_expand_big_items(rowptr, big, vis);
  vis.finish();
}

// Bottom-up step: one thread per vertex, early exit on the first parent found in the frontier
// bitmap (omp_beamer.cc:13-31).  A wave owns two bitmap words, so next/visited words are
// written whole, without atomics.
__global__ void __launch_bounds__(GDN_BLOCK)
bfs_bu_kernel(const eoff_t *__restrict__ in_rowptr, const vid_t *__restrict__ in_colidx, int32_t m,
              const unsigned *__restrict__ front, unsigned *__restrict__ next, unsigned *__restrict__ visited,
              int32_t *__restrict__ depth, int32_t next_level, BfsCounters *cnt) {
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned lane = gdn_lane();
  bool found = false;
  if (v < (unsigned)m) {
    const unsigned vw = visited[v >> 5];
    if (!((vw >> (v & 31)) & 1u)) {
      const eoff_t rb = in_rowptr[v], re = in_rowptr[v + 1];
      for (eoff_t k = rb; k < re; k++) {
        const vid_t u = in_colidx[k];
        if ((front[u >> 5] >> (u & 31)) & 1u) {
          found = true;
          break;
        }
      }
    }
  }
  if (found) depth[v] = next_level;
  const unsigned long long mask = __ballot(found);
  if ((lane & 31u) == 0 && v < (unsigned)m) {
    const unsigned bits = (unsigned)(mask >> (lane & 32u));
    next[v >> 5] = bits;
    if (bits) visited[v >> 5] |= bits;
  }
  if (lane == 0 && mask) atomicAdd(&cnt->awake, (unsigned long long)__popcll(mask));
}

__global__ void __launch_bounds__(GDN_BLOCK)
bfs_queue_to_bitmap(const vid_t *__restrict__ q, unsigned n, unsigned *__restrict__ bits) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < n) {
    const vid_t u = q[i];
    atomicOr(&bits[u >> 5], 1u << (u & 31));
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
bfs_bitmap_to_queue(const unsigned *__restrict__ bits, unsigned nwords, vid_t *__restrict__ q,
                    BfsCounters *cnt, unsigned cap) {
  const unsigned w = blockIdx.x * GDN_BLOCK + threadIdx.x;
  unsigned word = (w < nwords) ? bits[w] : 0u;
  const unsigned n = __popc(word);
  const unsigned incl = gdn_wave_incl_scan(n);
  const unsigned total = __shfl(incl, 63, 64);
  if (total == 0) return;
  unsigned base = 0;
  if (gdn_lane() == 63) base = atomicAdd(&cnt->next_count, total);
  base = __shfl(base, 63, 64);
  unsigned pos = base + incl - n;
  while (word) {
    const int b = __ffs((int)word) - 1;
    word &= word - 1u;
    if (pos < cap) q[pos] = (vid_t)(w * 32u + (unsigned)b);
    else cnt->overflow = 1u;
    pos++;
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

extern "C" {

int gdn_bfs_dev(const gdn_graph *g, const gdn_graph *gin, int32_t source, int32_t *d_dist, gdn_stats *stats) {
  GDN_REQUIRE(g != nullptr && d_dist != nullptr, "graph / d_dist");
  GDN_REQUIRE(source >= 0 && source < g->m, "source out of range");
  GDN_REQUIRE(gin == nullptr || gin->m == g->m, "in-CSR vertex count");
  const int32_t m = g->m;
  const unsigned nwords = ((unsigned)m + 31u) / 32u;
  const unsigned nwords_pad = (nwords + 63u) & ~63u;  // bottom-up waves write whole 2-word groups
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  HostTimer tprep, tsolve;
  tprep.start();
  DevBuf<unsigned> visited, front, next;
  DevBuf<vid_t> q0, q1;
  DevBuf<unsigned long long> bigitems;
  DevBuf<BfsCounters> cnt;
  const unsigned qcap = (unsigned)m;
  const uint64_t bigcap64 = g->nnz / EXP_CHUNK + (uint64_t)m / 64 + 1024;
  const unsigned bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
  GDN_TRY(visited.alloc(nwords_pad));
  GDN_TRY(q0.alloc(qcap));
  GDN_TRY(q1.alloc(qcap));
  GDN_TRY(bigitems.alloc(bigcap));
  GDN_TRY(cnt.alloc(1));
  if (gin) {
    GDN_TRY(front.alloc(nwords_pad));
    GDN_TRY(next.alloc(nwords_pad));
  }
  st.prep_ms = tprep.stop_ms();

  // ---- timed region == omp_beamer.cc:128-148 plus the depth initialisation
  tsolve.start();
  GDN_TRY(gdn_fill_i32(d_dist, GDN_MYINFINITY, (size_t)m, 0));
  GDN_HIP(hipMemsetAsync(visited.p, 0, (size_t)nwords_pad * 4, 0));
  hipLaunchKernelGGL(bfs_seed_kernel, dim3(1), dim3(64), 0, 0, source, d_dist, visited.p, q0.p);

  const int alpha = 15, beta = 18;
  vid_t *qin = q0.p, *qout = q1.p;
  unsigned nf = 1;
  int64_t edges_to_check = (int64_t)g->nnz;
  eoff_t srow[2];
  GDN_HIP(hipMemcpy(srow, g->rowptr + source, sizeof(srow), hipMemcpyDeviceToHost));
  int64_t scout_count = (int64_t)(srow[1] - srow[0]);
  int32_t level = 0;  // depth of the vertices in the current frontier
  int iter = 0;
  BfsCounters h;
  ExpBigList big;
  big.items = bigitems.p;
  big.capacity = bigcap;
  while (nf > 0) {
    if (gin != nullptr && scout_count > edges_to_check / alpha) {
      // ---- bottom-up phase (omp_beamer.cc:130-141)
      GDN_HIP(hipMemsetAsync(front.p, 0, (size_t)nwords_pad * 4, 0));
      hipLaunchKernelGGL(bfs_queue_to_bitmap, dim3(gdn_nblocks(nf)), dim3(GDN_BLOCK), 0, 0, qin, nf, front.p);
      int64_t awake = (int64_t)nf, old_awake;
      unsigned *fr = front.p, *nx = next.p;
      do {
        ++iter;
        old_awake = awake;
        GDN_HIP(hipMemsetAsync(cnt.p, 0, sizeof(BfsCounters), 0));
        hipLaunchKernelGGL(bfs_bu_kernel, dim3(gdn_nblocks((uint64_t)nwords_pad * 32)), dim3(GDN_BLOCK), 0, 0,
                           gin->rowptr, gin->colidx, m, fr, nx, visited.p, d_dist, level + 1, cnt.p);
        GDN_HIP(hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost));
        awake = (int64_t)h.awake;
        unsigned *t = fr;
        fr = nx;
        nx = t;
        level++;
      } while (awake >= old_awake || awake > m / beta);
      GDN_HIP(hipMemsetAsync(cnt.p, 0, sizeof(BfsCounters), 0));
      hipLaunchKernelGGL(bfs_bitmap_to_queue, dim3(gdn_nblocks(nwords)), dim3(GDN_BLOCK), 0, 0, fr, nwords, qin,
                         cnt.p, qcap);
      GDN_HIP(hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost));
      nf = h.next_count;
      scout_count = 1;
    } else {
      // ---- top-down step (omp_beamer.cc:143-146)
      ++iter;
      edges_to_check -= scout_count;
      GDN_HIP(hipMemsetAsync(cnt.p, 0, sizeof(BfsCounters), 0));
      BfsTdVis vis;
      vis.rowptr = g->rowptr;
      vis.colidx = g->colidx;
      vis.visited = visited.p;
      vis.depth = d_dist;
      vis.outq = qout;
      vis.cnt = cnt.p;
      vis.cap = qcap;
      vis.next_level = level + 1;
      vis.scout_local = 0;
      big.count = &cnt.p->big_count;
      big.overflow = &cnt.p->overflow;
      hipLaunchKernelGGL(bfs_td_kernel, dim3(gdn_nblocks(nf)), dim3(GDN_BLOCK), 0, 0, g->rowptr, qin, nf, big, vis);
      hipLaunchKernelGGL(bfs_td_big_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
      GDN_HIP(hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost));
      nf = h.next_count;
      scout_count = (int64_t)h.scout;
      vid_t *t = qin;
      qin = qout;
      qout = t;
      level++;
    }
    if (h.overflow) {
      gdn_set_error("gdn_bfs: device worklist overflow");
      return GDN_ERR_OVERFLOW;
    }
  }
  GDN_HIP(hipGetLastError());
  st.solve_ms = tsolve.stop_ms();
  st.iterations = iter;
  uint64_t te = 0;
  GDN_TRY(gdn_reached_edges(g, d_dist, GDN_MYINFINITY, &te));
  st.edges_traversed = te;
  if (stats) *stats = st;
  return GDN_OK;
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
