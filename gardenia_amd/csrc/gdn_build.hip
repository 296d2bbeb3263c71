// gdn_build.hip -- device-side graph construction (ingest row of SURVEY 8f + the synthetic
// input of SURVEY 8d).  NOT on the solver hot path.  The sorts are gdn_sort.hip's radix sort (no library primitive is
// left anywhere: round 1 called rocPRIM here).
//
//   gdn_rmat_build     : Graph500 R-MAT edge stream (include/generator.h:81-114; A=.57 B=.19
//                        C=.19) from the counter-based RNG specified in
//                        gardenia_amd/graphio.py (bit-identical to rmat_edges there), then
//                        the clean-up the reference loader applies: self loops dropped
//                        (csr_graph.h:108), neighbour lists ascending (:127), duplicates
//                        dropped (:132-143).
//   gdn_graph_transpose: reverse graph (csr_graph.h:170-194), rows ascending.
// The reference builds both serially with vector<vector<int>> and an O(deg^2) erase loop.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <string.h>
#include <stdlib.h>


#include "gdn_expand.hpp"
#include "gdn_pb.hpp"

#define RMAT_TA 2448131358u   // int(0.57 * 2^32)
#define RMAT_TAB 3264175144u  // int(0.76 * 2^32)
#define RMAT_TABC 4080218931u // int(0.95 * 2^32)

__device__ __forceinline__ unsigned long long rmat_mix64(unsigned long long z) {
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}

__device__ __forceinline__ unsigned long long rmat_permute(unsigned long long x, int scale,
                                                           unsigned long long seed) {
  const unsigned long long mask = (1ull << scale) - 1ull;
  const int half = scale / 2 > 1 ? scale / 2 : 1;
  x &= mask;
  x = (x * 0x9E3779B1ull + seed) & mask;
  x ^= x >> half;
  x = (x * 0x85EBCA6Bull + 0xC2B2AE35ull) & mask;
  x ^= x >> half;
  x = (x * 0x27D4EB2Full + 0x165667B1ull) & mask;
  x ^= x >> half;
  return x;
}

// edge number e of the stream: scale levels of the quadrant choice, 32 random bits each (two levels per 64-bit mix)
__device__ __forceinline__ void rmat_edge(int scale, unsigned long long e, unsigned long long seed, int permute, unsigned t_a,
                                          unsigned t_ab, unsigned t_abc, unsigned long long &src, unsigned long long &dst) {
  const unsigned long long base = seed + e * 0x9E3779B97F4A7C15ull;
  unsigned long long h = 0;
  src = 0;
  dst = 0;
  for (int l = 0; l < scale; l++) {
    unsigned r;
    if ((l & 1) == 0) {
      h = rmat_mix64(base + (unsigned long long)(l >> 1) * 0xBF58476D1CE4E5B9ull);
      r = (unsigned)(h & 0xFFFFFFFFull);
    } else {
      r = (unsigned)(h >> 32);
    }
    src <<= 1;
    dst <<= 1;
    if (r >= t_abc) {
      src |= 1ull;
      dst |= 1ull;
    } else if (r >= t_ab) {
      src |= 1ull;
    } else if (r >= t_a) {
      dst |= 1ull;
    }
  }
  if (permute) {
    src = rmat_permute(src, scale, seed);
    dst = rmat_permute(dst, scale, seed);
  }
}
// key = (row << 32) | col ; by_dst: row = dst (in-CSR) else row = src (out-CSR)
__global__ void __launch_bounds__(GDN_BLOCK)
rmat_keys_kernel(int scale, unsigned long long nedges, unsigned long long seed, int permute, int by_dst,
                 unsigned long long *__restrict__ keys, unsigned t_a = RMAT_TA, unsigned t_ab = RMAT_TAB,
                 unsigned t_abc = RMAT_TABC) {
  unsigned long long e = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  for (; e < nedges; e += stride) {
    unsigned long long src, dst;
    rmat_edge(scale, e, seed, permute, t_a, t_ab, t_abc, src, dst);
    keys[e] = by_dst ? ((dst << 32) | src) : ((src << 32) | dst);
  }
}
// gdn_rmat_build_range: the WHOLE stream is generated, only the in-CSR keys (dst << 32 | src) of the destinations [v_lo, v_hi) are
// kept -- appended in any order (one reservation per wave; the sort that follows orders them), self loops left out.  keys ==
// nullptr: count only (the first pass sizes the buffer of the second).
__global__ void __launch_bounds__(GDN_BLOCK)
rmat_range_keys_kernel(int scale, unsigned long long nedges, unsigned long long seed, int permute, unsigned t_a, unsigned t_ab,
                       unsigned t_abc, unsigned v_lo, unsigned v_hi, unsigned long long *__restrict__ keys,
                       unsigned long long *__restrict__ counter, unsigned long long capacity) {
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  const unsigned long long rounds = (nedges + stride - 1) / stride;  // (whole waves stay together for the ballots)
  unsigned long long e = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  unsigned long long mine = 0;
  for (unsigned long long it = 0; it < rounds; it++, e += stride) {
    unsigned long long src = 0, dst = 0;
    bool keep = false;
    if (e < nedges) {
      rmat_edge(scale, e, seed, permute, t_a, t_ab, t_abc, src, dst);
      keep = dst >= v_lo && dst < v_hi && src != dst;
    }
    if (keys == nullptr) {
      mine += keep ? 1ull : 0ull;
      continue;
    }
    const unsigned long long mask = __ballot(keep);
    if (mask == 0ull) continue;
    unsigned long long at = 0;
    if (gdn_lane() == (unsigned)(__ffsll((long long)mask) - 1)) at = atomicAdd(counter, (unsigned long long)__popcll(mask));
    at = __shfl(at, __ffsll((long long)mask) - 1, 64) + (unsigned long long)__popcll(mask & gdn_lanemask_lt());
    if (keep && at < capacity) keys[at] = (dst << 32) | src;
  }
  if (keys == nullptr) {
    mine = gdn_wave_sum(mine);
    if (gdn_lane() == 0 && mine) atomicAdd(counter, mine);
  }
}
// out-degree contributions of a set of (unique) in-edges: deg[src] += 1
__global__ void __launch_bounds__(GDN_BLOCK)
out_degree_add_kernel(const vid_t *__restrict__ colidx, unsigned long long nnz, int32_t *__restrict__ deg) {
  unsigned long long k = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  for (; k < nnz; k += stride) atomicAdd(&deg[colidx[k]], 1);
}

// ---- gdn_rmat_build_ex, GDN_RMAT_COMPACT: the ids that occur in no edge (self loops do not count) are dropped and the
// others renumbered in ascending order -- "few isolated vertices", like the real graphs of BASELINE configs 2 and 4
__global__ void __launch_bounds__(GDN_BLOCK)
rmat_mark_ids_kernel(const unsigned long long *__restrict__ keys, unsigned long long n, unsigned *__restrict__ bits) {
  unsigned long long i = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  for (; i < n; i += stride) {
    const unsigned a = (unsigned)(keys[i] >> 32), b = (unsigned)(keys[i] & 0xFFFFFFFFull);
    if (a == b) continue;
    if (!((bits[a >> 5] >> (a & 31u)) & 1u)) atomicOr(&bits[a >> 5], 1u << (a & 31u));
    if (!((bits[b >> 5] >> (b & 31u)) & 1u)) atomicOr(&bits[b >> 5], 1u << (b & 31u));
  }
}
__global__ void __launch_bounds__(GDN_BLOCK)
rmat_word_counts_kernel(const unsigned *__restrict__ bits, unsigned nwords, unsigned *__restrict__ cnt) {
  const unsigned w = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (w < nwords) cnt[w] = (unsigned)__popc(bits[w]);
}
__global__ void __launch_bounds__(GDN_BLOCK)
rmat_relabel_kernel(unsigned long long *__restrict__ keys, unsigned long long n, const unsigned *__restrict__ bits,
                    const eoff_t *__restrict__ word_base) {
  unsigned long long i = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  for (; i < n; i += stride) {
    const unsigned a = (unsigned)(keys[i] >> 32), b = (unsigned)(keys[i] & 0xFFFFFFFFull);
    if (a == b) {  // a self loop (dropped by csr_from_keys): any id inside the new range
      keys[i] = 0ull;
      continue;
    }
    const unsigned na = (unsigned)word_base[a >> 5] + (unsigned)__popc(bits[a >> 5] & ((1u << (a & 31u)) - 1u));
    const unsigned nb = (unsigned)word_base[b >> 5] + (unsigned)__popc(bits[b >> 5] & ((1u << (b & 31u)) - 1u));
    keys[i] = ((unsigned long long)na << 32) | nb;
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
keys_flag_kernel(const unsigned long long *__restrict__ keys, unsigned long long n, unsigned *__restrict__ flag) {
  unsigned long long i = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  for (; i < n; i += stride) {
    const unsigned long long k = keys[i];
    const bool self = (unsigned)(k >> 32) == (unsigned)(k & 0xFFFFFFFFull);
    const bool dup = (i > 0) && keys[i - 1] == k;
    flag[i] = (!self && !dup) ? 1u : 0u;
  }
}

// Rows without a key take the offset of the next row that has one.  The thread of a row's first key fills the rows in
// front of it -- a few, on the graphs of the bench; but a run of empty rows is ONE thread's loop, and the rank-ordered DAG
// of the forward triangle count starts with every vertex of degree 0 (half of an R-MAT graph): 4 M stores by one thread,
// 100 ms per CSR, 200 of the 240 ms of that plan build (profiles/sessions/r04_103.sh).  Runs of more than KEYS_GAP_LONG
// rows are listed instead and filled by the whole grid (keys_fill_gaps_kernel).
#define KEYS_GAP_LONG 256
struct KeysGap {
  unsigned long long first, last;  // rows [first, last]
  eoff_t value;
};
__global__ void __launch_bounds__(GDN_BLOCK)
keys_compact_kernel(const unsigned long long *__restrict__ keys, const unsigned *__restrict__ flag,
                    const eoff_t *__restrict__ pos, unsigned long long n, int32_t m, vid_t *__restrict__ colidx,
                    eoff_t *__restrict__ rowptr, KeysGap *__restrict__ gaps, unsigned *__restrict__ n_gaps) {
  unsigned long long i = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  for (; i < n; i += stride) {
    const unsigned long long k = keys[i];
    if (flag[i]) colidx[pos[i]] = (vid_t)(unsigned)(k & 0xFFFFFFFFull);
    // row boundaries on the unfiltered sorted stream: rowptr[r] = #kept before the first key of row >= r
    const long long row = (long long)(k >> 32);
    const long long prev = (i > 0) ? (long long)(keys[i - 1] >> 32) : -1ll;
    if (row - prev > KEYS_GAP_LONG) {
      KeysGap g;
      g.first = (unsigned long long)(prev + 1);
      g.last = (unsigned long long)row;
      g.value = pos[i];
      gaps[atomicAdd(n_gaps, 1u)] = g;  // (at most m / KEYS_GAP_LONG + 1 of them: the list's capacity)
    } else {
      for (long long r = prev + 1; r <= row; r++) rowptr[r] = pos[i];
    }
    if (i + 1 == n && row < (long long)m) {  // the rows behind the last key
      KeysGap g;
      g.first = (unsigned long long)(row + 1);
      g.last = (unsigned long long)m;
      g.value = pos[n];
      gaps[atomicAdd(n_gaps, 1u)] = g;
    }
  }
}
__global__ void __launch_bounds__(GDN_BLOCK)
keys_fill_gaps_kernel(const KeysGap *__restrict__ gaps, const unsigned *__restrict__ n_gaps, eoff_t *__restrict__ rowptr) {
  const unsigned long long tid = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  const unsigned ng = *n_gaps;
  for (unsigned g = 0; g < ng; g++) {
    const KeysGap gp = gaps[g];
    for (unsigned long long r = gp.first + tid; r <= gp.last; r += stride) rowptr[r] = gp.value;
  }
}

__global__ void __launch_bounds__(GDN_BLOCK) zero_rowptr_kernel(eoff_t *rowptr, int32_t m) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i <= (unsigned)m) rowptr[i] = 0;
}

static int bits_for(int32_t m) {
  int b = 1;
  while (b < 32 && (1ll << b) < (long long)m) b++;
  return b;
}

// stable radix sort on the bits [begin_bit, bits) of the keys (gdn_sort.hip); returns the buffer holding the result and
// frees the other
int gdn_radix_sort_u64(unsigned long long *a, unsigned long long *b, unsigned long long n, unsigned begin_bit, unsigned end_bit,
                       const unsigned long long **sorted);

static int sort_keys(DevBuf<unsigned long long> &ka, DevBuf<unsigned long long> &kb, unsigned long long n,
                     unsigned bits, const unsigned long long **sorted_out, bool keep_both = false, unsigned begin_bit = 0) {
  if (bits > 64) bits = 64;
  const unsigned long long *sorted = nullptr;
  GDN_TRY(gdn_radix_sort_u64(ka.p, kb.p, n, begin_bit, bits, &sorted));
  // free the non-current key buffer early (unless the buffers are a plan's scratch, reused by its next build)
  if (!keep_both) {
    if (sorted == ka.p) kb.release();
    else ka.release();
  }
  *sorted_out = sorted;
  return GDN_OK;
}

// sorted 64-bit keys (row<<32|col) -> owned CSR graph; drops self loops and duplicates
static int csr_from_keys(DevBuf<unsigned long long> &ka, DevBuf<unsigned long long> &kb, unsigned long long n,
                         int32_t m, int row_bits, gdn_graph **out) {
  const unsigned long long *sorted = nullptr;
  GDN_TRY(sort_keys(ka, kb, n, (unsigned)(32 + row_bits), &sorted));
  hipError_t e;
  DevBuf<unsigned> flag;
  DevBuf<eoff_t> pos;
  GDN_TRY(flag.alloc_scratch(n));
  GDN_TRY(pos.alloc_scratch(n + 1));
  unsigned nb = (unsigned)((n + GDN_BLOCK - 1) / GDN_BLOCK > 262144ull ? 262144ull : (n + GDN_BLOCK - 1) / GDN_BLOCK);
  if (nb == 0) nb = 1;
  hipLaunchKernelGGL(keys_flag_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, sorted, n, flag.p);
  GDN_HIP(hipGetLastError());
  GDN_TRY(gdn_exclusive_scan_u32_to_u64(flag.p, pos.p, (size_t)n, 0));
  eoff_t nnz = 0;
  GDN_HIP(hipMemcpy(&nnz, pos.p + n, sizeof(eoff_t), hipMemcpyDeviceToHost));
  gdn_graph *g = new gdn_graph();
  g->m = m;
  g->nnz = nnz;
  g->owned = true;
  e = gdn_plain_malloc((void **)&g->rowptr, ((size_t)m + 1) * sizeof(eoff_t));
  if (e == hipSuccess) e = gdn_plain_malloc((void **)&g->colidx, (nnz ? nnz : 1) * sizeof(vid_t));
  if (e != hipSuccess) {
    gdn_set_error("csr_from_keys: %s", hipGetErrorString(e));
    gdn_graph_free(g);
    return GDN_ERR_OOM;
  }
  if (n == 0) {
    hipLaunchKernelGGL(zero_rowptr_kernel, dim3(gdn_nblocks((uint64_t)m + 1)), dim3(GDN_BLOCK), 0, 0, g->rowptr, m);
  } else {
    DevBuf<KeysGap> gaps;
    DevBuf<unsigned> n_gaps;
    int rc = gaps.alloc_scratch((size_t)m / KEYS_GAP_LONG + 2);
    if (rc == GDN_OK) rc = n_gaps.alloc_scratch(1);
    if (rc != GDN_OK || hipMemsetAsync(n_gaps.p, 0, 4, 0) != hipSuccess) {
      gdn_graph_free(g);
      return rc != GDN_OK ? rc : GDN_ERR_HIP;
    }
    hipLaunchKernelGGL(keys_compact_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, sorted, flag.p, pos.p, n, m, g->colidx,
                       g->rowptr, gaps.p, n_gaps.p);
    hipLaunchKernelGGL(keys_fill_gaps_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, gaps.p, n_gaps.p, g->rowptr);
    GDN_HIP(hipGetLastError());
    GDN_HIP(hipDeviceSynchronize());  // (gaps / n_gaps go back to the scratch cache here)
  }
  GDN_HIP(hipGetLastError());
  GDN_HIP(hipDeviceSynchronize());
  *out = g;
  return GDN_OK;
}

// sorted-or-not 64-bit keys (row << 32 | col) of a graph on m vertices -> owned CSR (self loops and duplicates dropped, rows
// ascending); the key buffers are consumed.  For builders in other translation units (gdn_tc.hip's rank-ordered DAG).
int gdn_build_csr_from_keys(DevBuf<unsigned long long> &ka, DevBuf<unsigned long long> &kb, unsigned long long n, int32_t m,
                            gdn_graph **out) {
  return csr_from_keys(ka, kb, n, m, bits_for(m), out);
}

struct KeyVis {
  const vid_t *__restrict__ colidx;
  unsigned long long *__restrict__ keys;
  int32_t v;
  unsigned long long *__restrict__ fwd = nullptr;  // optional second array: the untransposed key (row<<32 | col)
  __device__ __forceinline__ void begin_big(vid_t vv) { v = vv; }
  __device__ __forceinline__ void edge(int owner, eoff_t k, bool valid) {
    const unsigned src = (unsigned)__shfl(v, owner, 64);
    if (valid) {
      const unsigned col = (unsigned)colidx[k];
      keys[k] = ((unsigned long long)col << 32) | src;
      if (fwd) fwd[k] = ((unsigned long long)src << 32) | col;
    }
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
transpose_keys_kernel(const eoff_t *__restrict__ rowptr, int32_t m, ExpBigList big, KeyVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vis.v = (int32_t)v;
  if (v < (unsigned)m) {
    b = rowptr[v];
    e = rowptr[v + 1];
  }
  gdn_expand_wave(b, e, (vid_t)v, big, vis, s_scan[threadIdx.x >> 6]);
}

__global__ void __launch_bounds__(GDN_BLOCK)
transpose_keys_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, KeyVis vis) {
  vis.v = 0;
  gdn_expand_big_items(rowptr, big, vis);
}

// ------------------------------------------------------------------------------------------
// propagation-blocking layout (gdn_pb.hpp): edges sorted by (source chunk, destination bin,
// destination row, source id)
// ------------------------------------------------------------------------------------------
struct PbKeyVis {
  const vid_t *__restrict__ colidx;
  unsigned long long *__restrict__ keys;
  const eoff_t *__restrict__ cs;  // compact source index of every global id (nullptr = identity)
  const eoff_t *__restrict__ cd;  // compact row index of every local row (nullptr = identity)
  int log_chunk, log_bin;
  int bin_bits;
  int transposed;  // the CSR's rows are the SOURCES (out-CSR): swap the roles
  int32_t v;
  // source classes (PageRank hub tier): only edges whose source has class `want` belong to this layout, the
  // others get the sentinel key (sorts behind every real key).  src_major: tile order (source, row) instead of
  // (row, source).  nvalid: per-lane count of the edges that got a real key.
  const uint8_t *__restrict__ cls = nullptr;
  int want = 0;
  const uint8_t *__restrict__ dcls = nullptr;  // row classes (original row ids), same convention as cls
  int dwant = 0;
  int src_major = 0;
  unsigned long long sentinel = 0;
  unsigned long long nvalid = 0;
  unsigned long long *nvalid_out = nullptr;
  __device__ __forceinline__ void begin_big(vid_t vv) { v = vv; }
  __device__ __forceinline__ void edge(int owner, eoff_t k, bool valid) {
    unsigned row = (unsigned)__shfl(v, owner, 64);
    if (valid) {
      unsigned col = (unsigned)colidx[k];
      if (transposed) {
        const unsigned t = row;
        row = col;
        col = t;
      }
      if ((cls && (int)cls[col] != want) || (dcls && (int)dcls[row] != dwant)) {
        keys[k] = sentinel;
        return;
      }
      nvalid++;
      if (cs) col = (unsigned)cs[col];
      if (cd) row = (unsigned)cd[row];
      const unsigned long long chunk = col >> log_chunk, bin = row >> log_bin;
      const unsigned long long vl = row & ((1u << log_bin) - 1u), ul = col & ((1u << log_chunk) - 1u);
      const unsigned long long in_tile = src_major ? ((ul << log_bin) | vl) : ((vl << log_chunk) | ul);
      keys[k] = (chunk << (bin_bits + log_bin + log_chunk)) | (bin << (log_bin + log_chunk)) | in_tile;
    }
  }
  __device__ __forceinline__ void finish() {
    if (!nvalid_out) return;
    const unsigned long long s = gdn_wave_sum(nvalid);
    if (gdn_lane() == 0 && s) atomicAdd(nvalid_out, s);
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
pb_keys_kernel(const eoff_t *__restrict__ rowptr, int32_t m, ExpBigList big, PbKeyVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vis.v = (int32_t)v;
  if (v < (unsigned)m) {
    b = rowptr[v];
    e = rowptr[v + 1];
  }
  gdn_expand_wave(b, e, (vid_t)v, big, vis, s_scan[threadIdx.x >> 6]);
  vis.finish();
}

__global__ void __launch_bounds__(GDN_BLOCK)
pb_keys_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, PbKeyVis vis) {
  vis.v = 0;
  gdn_expand_big_items(rowptr, big, vis);
  vis.finish();
}

// ---- vertex compaction helpers
struct PbMarkVis {
  const vid_t *__restrict__ colidx;
  uint32_t *__restrict__ mark;
  const uint8_t *__restrict__ cls = nullptr;  // source classes: only sources of class `want` are marked
  int want = 0;
  __device__ __forceinline__ void begin_big(vid_t) {}
  __device__ __forceinline__ void edge(int, eoff_t k, bool valid) {
    if (valid) {
      const vid_t c = colidx[k];
      if (!cls || (int)cls[c] == want) mark[c] = 1u;  // benign race: everybody stores 1
    }
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
pb_mark_kernel(const eoff_t *__restrict__ rowptr, int32_t m, ExpBigList big, PbMarkVis vis, uint32_t *__restrict__ dflag,
               const uint8_t *__restrict__ dcls, int dwant, int rows_of_class_only) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  if (v < (unsigned)m) {
    b = rowptr[v];
    e = rowptr[v + 1];
    // rows_of_class_only: the row slices hold the rows of class dwant only (hub-row layout); otherwise every row
    // with entries, whatever its class (the bins of the main and hub-source layouts must be the same)
    dflag[v] = (e > b && (!rows_of_class_only || (int)dcls[v] == dwant)) ? 1u : 0u;
  }
  gdn_expand_wave(b, e, (vid_t)v, big, vis, s_scan[threadIdx.x >> 6]);
}

__global__ void __launch_bounds__(GDN_BLOCK)
pb_mark_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, PbMarkVis vis) {
  gdn_expand_big_items(rowptr, big, vis);
}

// compact index i -> (i / per) * 2^lg + i % per: `per` vertices per slice, slices still 2^lg apart
__global__ void __launch_bounds__(GDN_BLOCK)
pb_respace_kernel(eoff_t *__restrict__ cidx, size_t n, uint64_t per, int lg) {
  const size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < n) {
    const eoff_t c = cidx[i];
    cidx[i] = ((c / per) << lg) + (c % per);
  }
}

// inverse of the compact index over the active vertices
__global__ void __launch_bounds__(GDN_BLOCK)
pb_inverse_kernel(const uint32_t *__restrict__ flag, const eoff_t *__restrict__ cidx, size_t n, uint32_t *__restrict__ inv) {
  const size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < n && flag[i]) inv[cidx[i]] = (uint32_t)i;
}

// activity bitmap + original id of the first active vertex of every slice
__global__ void __launch_bounds__(GDN_BLOCK)
pb_slices_kernel(const uint32_t *__restrict__ flag, const eoff_t *__restrict__ cidx, unsigned n, int log_slice,
                 unsigned nslices, uint32_t *__restrict__ bits, uint32_t *__restrict__ lo) {
  const unsigned w = blockIdx.x * GDN_BLOCK + threadIdx.x;  // one 32-id word per thread
  if (w > ((n + 31u) >> 5)) return;
  unsigned word = 0;
  for (unsigned i = 0; i < 32; i++) {
    const unsigned id = (w << 5) + i;
    if (id < n && flag[id]) {
      word |= 1u << i;
      const eoff_t c = cidx[id];
      if ((c & ((1ull << log_slice) - 1ull)) == 0 && (c >> log_slice) > 0) lo[c >> log_slice] = id;
    }
  }
  if (w < ((n + 31u) >> 5)) bits[w] = word;
  if (w == 0) {
    lo[0] = 0;  // slice 0 also owns the inactive ids in front of its first active one
    lo[nslices] = n;
  }
}

// tsu[t] = index of the first sorted key whose tile id (chunk*nbins + bin) is >= t
__global__ void __launch_bounds__(GDN_BLOCK)
pb_bounds_kernel(const unsigned long long *__restrict__ keys, unsigned long long n, int shift, int bin_bits,
                 unsigned nbins, unsigned long long ntiles, eoff_t *__restrict__ tsu) {
  unsigned long long i = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  const unsigned long long bmask = (1ull << bin_bits) - 1ull;
  for (; i < n; i += stride) {
    const unsigned long long k = keys[i] >> shift;
    const long long t = (long long)((k >> bin_bits) * nbins + (k & bmask));
    long long tp = -1;
    if (i > 0) {
      const unsigned long long kp = keys[i - 1] >> shift;
      tp = (long long)((kp >> bin_bits) * nbins + (kp & bmask));
    }
    for (long long x = tp + 1; x <= t; x++) tsu[x] = i;
    if (i + 1 == n)
      for (long long x = t + 1; x <= (long long)ntiles; x++) tsu[x] = n;
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
pb_fill_u64_kernel(eoff_t *p, unsigned long long n, eoff_t v) {
  unsigned long long i = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void __launch_bounds__(GDN_BLOCK)
pb_fill_u16_kernel(uint16_t *p, unsigned long long n, uint16_t v) {
  unsigned long long i = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  for (; i < n; i += stride) p[i] = v;
}

// padded tile sizes in chunk-major (psz_c) and bin-major (psz_b) tile order
__global__ void __launch_bounds__(GDN_BLOCK)
pb_tile_sizes_kernel(const eoff_t *__restrict__ tsu, unsigned nchunks, unsigned nbins, unsigned pad,
                     uint32_t *__restrict__ psz_c, uint32_t *__restrict__ psz_b, const eoff_t *__restrict__ ds) {
  const unsigned long long t = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (t >= (unsigned long long)nchunks * nbins) return;
  const unsigned c = (unsigned)(t / nbins), b = (unsigned)(t % nbins);
  const eoff_t fill = ds ? ds[tsu[t + 1]] - ds[tsu[t]] : 0;  // filler edges of the delta-coded row stream
  const uint32_t sz = (uint32_t)(((tsu[t + 1] - tsu[t]) + fill + (pad - 1)) & ~(eoff_t)(pad - 1));
  psz_c[t] = sz;
  psz_b[(unsigned long long)b * nchunks + c] = sz;
}

__global__ void __launch_bounds__(GDN_BLOCK)
pb_ptrs_kernel(const eoff_t *__restrict__ pu, const eoff_t *__restrict__ pv, unsigned nchunks, unsigned nbins,
               eoff_t *__restrict__ chunk_ptr, eoff_t *__restrict__ bin_ptr) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i <= nchunks) chunk_ptr[i] = pu[(unsigned long long)i * nbins];
  if (i <= nbins) bin_ptr[i] = pv[(unsigned long long)i * nchunks];
}

// slice starts are moved to aligned positions: every tile of chunk c (bin b) shifts by du[c] (dv[b])
__global__ void __launch_bounds__(GDN_BLOCK)
pb_shift_kernel(eoff_t *__restrict__ pu, eoff_t *__restrict__ pv, const eoff_t *__restrict__ du,
                const eoff_t *__restrict__ dv, unsigned nchunks, unsigned nbins) {
  const unsigned long long t = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (t >= (unsigned long long)nchunks * nbins) return;
  pu[t] += du[t / nbins];
  pv[t] += dv[t / nchunks];
}

__global__ void __launch_bounds__(GDN_BLOCK)
pb_fill_u32_kernel(uint32_t *p, unsigned long long n, uint32_t v) {
  unsigned long long i = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  for (; i < n; i += stride) p[i] = v;
}

// ---- 8-bit delta coding of the row stream V (PbPlan::v8): inside a tile the edges are sorted by row, so the row of
// an edge is stored as its distance to the previous edge of the same 32-edge group (+ one u16 base per group).
// nd[i] = filler edges (source = the pad id, value 0) in front of sorted key i so that no distance exceeds 255.
__global__ void __launch_bounds__(GDN_BLOCK)
pb_gap_kernel(const unsigned long long *__restrict__ keys, unsigned long long n, int log_chunk, int log_bin,
              uint32_t *__restrict__ nd) {
  unsigned long long i = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  for (; i < n; i += stride) {
    uint32_t f = 0;
    if (i > 0) {
      const unsigned long long k = keys[i], kp = keys[i - 1];
      if ((k >> (log_chunk + log_bin)) == (kp >> (log_chunk + log_bin))) {
        const unsigned r = (unsigned)(k >> log_chunk) & ((1u << log_bin) - 1u);
        const unsigned rp = (unsigned)(kp >> log_chunk) & ((1u << log_bin) - 1u);
        const unsigned gap = r - rp;
        if (gap > 255u) f = (gap - 1u) / 255u;
      }
    }
    nd[i] = f;
  }
}

// pads behind the edges of a tile repeat its last row (distance 0) instead of row 0
__global__ void __launch_bounds__(GDN_BLOCK)
pb_pad_rows_kernel(const eoff_t *__restrict__ tsu, const eoff_t *__restrict__ ds, const eoff_t *__restrict__ pv,
                   const uint32_t *__restrict__ psz_b, unsigned nchunks, unsigned nbins, uint16_t *__restrict__ V) {
  const unsigned long long t = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (t >= (unsigned long long)nchunks * nbins) return;
  const unsigned c = (unsigned)(t / nbins), b = (unsigned)(t % nbins);
  const eoff_t cnt = (tsu[t + 1] - tsu[t]) + (ds[tsu[t + 1]] - ds[tsu[t]]);
  if (cnt == 0) return;
  const unsigned long long tb = (unsigned long long)b * nchunks + c;
  const eoff_t base = pv[tb];
  const uint16_t last = V[base + cnt - 1];
  for (eoff_t j = cnt; j < (eoff_t)psz_b[tb]; j++) V[base + j] = last;
}

__global__ void __launch_bounds__(GDN_BLOCK)
pb_encode_rows_kernel(const uint16_t *__restrict__ V, unsigned long long n, uint8_t *__restrict__ Vd,
                      uint16_t *__restrict__ Vb, unsigned *__restrict__ bad) {
  unsigned long long i = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  for (; i < n; i += stride) {
    const unsigned v = V[i];
    unsigned d = 0;
    if (i & 31ull) {
      const unsigned vp = V[i - 1];
      d = v - vp;
      if (v < vp || d > 255u) {
        *bad = 1u;
        d = 0;
      }
    } else {
      Vb[i >> 5] = (uint16_t)v;
    }
    Vd[i] = (uint8_t)d;
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
pb_scatter_kernel(const unsigned long long *__restrict__ keys, unsigned long long n, int log_chunk, int log_bin,
                  int bin_bits, unsigned nchunks, unsigned nbins, const eoff_t *__restrict__ tsu,
                  const eoff_t *__restrict__ pu, const eoff_t *__restrict__ pv, uint16_t *__restrict__ U,
                  uint16_t *__restrict__ V, const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx,
                  const float *__restrict__ ev_in, float *__restrict__ ev_out, int randv, int transposed,
                  int src_major, const uint32_t *__restrict__ nd, const eoff_t *__restrict__ ds,
                  const uint32_t *__restrict__ inv_s, const uint32_t *__restrict__ inv_d) {
  unsigned long long i = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  const unsigned long long bmask = (1ull << bin_bits) - 1ull;
  for (; i < n; i += stride) {
    const unsigned long long k = keys[i];
    const unsigned long long cb = k >> (log_chunk + log_bin);
    const unsigned long long b = cb & bmask, c = cb >> bin_bits;
    const unsigned long long t = c * nbins + b;
    unsigned long long off = i - tsu[t];
    if (ds) {  // delta-coded rows: nd[i] fillers sit in front of this edge, 255 rows apart behind the previous edge
      const unsigned f = nd[i];
      off += (ds[i] - ds[tsu[t]]) + f;
      if (f) {
        const unsigned rp = (unsigned)(keys[i - 1] >> log_chunk) & ((1u << log_bin) - 1u);
        for (unsigned j = 0; j < f; j++)  // U keeps the pad id (value 0) from the fill
          V[pv[b * nchunks + c] + off - f + j] = (uint16_t)(rp + 255u * (j + 1u));
      }
    }
    const unsigned ul = src_major ? (unsigned)(k >> log_bin) & ((1u << log_chunk) - 1u) : (unsigned)k & ((1u << log_chunk) - 1u);
    const unsigned vl = src_major ? (unsigned)k & ((1u << log_bin) - 1u) : (unsigned)(k >> log_chunk) & ((1u << log_bin) - 1u);
    U[pu[t] + off] = (uint16_t)ul;
    V[pv[b * nchunks + c] + off] = randv ? (uint16_t)((i * 2654435761ull >> 7) & ((1u << log_bin) - 1u)) : (uint16_t)vl;
    if (ev_in) {  // value of this edge: find the column in its (ascending) CSR row
      unsigned long long row = (b << log_bin) + vl;  // destination
      vid_t col = (vid_t)((c << log_chunk) + ul);    // source
      if (inv_s) {  // compacted layout: back to the original ids
        row = inv_d[row];
        col = (vid_t)inv_s[(size_t)col];
      }
      if (transposed) {                               // out-CSR: row = source, column = destination
        const unsigned long long t = row;
        row = (unsigned long long)col;
        col = (vid_t)t;
      }
      eoff_t lo = rowptr[row], hi = rowptr[row + 1];
      while (lo < hi) {
        const eoff_t mid = lo + ((hi - lo) >> 1);
        if (colidx[mid] < col) lo = mid + 1;
        else hi = mid;
      }
      ev_out[pu[t] + off] = ev_in[lo];
    }
  }
}

// group table: the q-th group of 2^log_group edges of tile t in chunk-major order -> its group index in bin-major
__global__ void __launch_bounds__(GDN_BLOCK)
pb_groups_kernel(const eoff_t *__restrict__ pu, const eoff_t *__restrict__ pv, const uint32_t *__restrict__ psz_c,
                 unsigned nchunks, unsigned nbins, uint32_t *__restrict__ G, int identity, int log_group) {
  const unsigned long long t = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (t >= (unsigned long long)nchunks * nbins) return;
  const unsigned c = (unsigned)(t / nbins), b = (unsigned)(t % nbins);
  const eoff_t gu = pu[t] >> log_group, ng = psz_c[t] >> log_group;
  const eoff_t gv = pv[(unsigned long long)b * nchunks + c] >> log_group;
  for (eoff_t q = 0; q < ng; q++) G[gu + q] = identity ? (uint32_t)(gu + q) : (uint32_t)(gv + q);
}

// ---- hub selection: out-edge counts of the sources from a 1/16 sample of the rows (any classification is
// correct, it only decides which edges take the cheap path), a log2 histogram of them, then the class flags
#define PB_HUB_SAMPLE_LOG 4
__global__ void __launch_bounds__(GDN_BLOCK)
pb_hub_sample_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, int32_t m,
                     uint32_t *__restrict__ cnt) {
  // one wave per sampled row
  const unsigned wid = (blockIdx.x * GDN_BLOCK + threadIdx.x) >> 6;
  const uint64_t row = (uint64_t)wid << PB_HUB_SAMPLE_LOG;
  if (row >= (uint64_t)m) return;
  const eoff_t b = rowptr[row], e = rowptr[row + 1];
  for (eoff_t k = b + gdn_lane(); k < e; k += 64) atomicAdd(&cnt[colidx[k]], 1u);
}

// histogram buckets: 4 per octave; bucket(c) = 4*floor(log2 c) + the two bits below the leading one
#define PB_HUB_BUCKETS 128
__host__ __device__ static inline unsigned pb_hub_bucket(unsigned c) {
  unsigned l = 0;
  while ((c >> l) > 1u) l++;
  return 4u * l + (l >= 2 ? ((c >> (l - 2)) & 3u) : 0u);
}
static inline unsigned pb_hub_bucket_floor(unsigned b) {  // smallest count that falls into bucket b (l >= 2)
  const unsigned l = b >> 2, f = b & 3u;
  return l >= 2 ? (4u + f) << (l - 2) : (1u << l);
}

__global__ void __launch_bounds__(GDN_BLOCK)
pb_hub_hist_kernel(const uint32_t *__restrict__ cnt, size_t n, unsigned *__restrict__ hist /*PB_HUB_BUCKETS*/) {
  __shared__ unsigned s_h[PB_HUB_BUCKETS];
  if (threadIdx.x < PB_HUB_BUCKETS) s_h[threadIdx.x] = 0;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * GDN_BLOCK) {
    const unsigned c = cnt[i];
    if (c) atomicAdd(&s_h[pb_hub_bucket(c)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < PB_HUB_BUCKETS && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}

// the same counts as a LINEAR histogram (one bin per count, counts >= PB_LIN_BINS - 1 in the last bin): the mid tiers
// are cut at arbitrary counts, not at bucket floors -- a tier holds at most PB_MID_MAX sources, and below the hubs one
// quarter-octave bucket of an R-MAT graph holds more than that
#define PB_LIN_BINS 4096
__global__ void __launch_bounds__(GDN_BLOCK)
pb_hub_hist_lin_kernel(const uint32_t *__restrict__ cnt, size_t n, unsigned *__restrict__ hist /*PB_LIN_BINS*/) {
  __shared__ unsigned s_h[PB_LIN_BINS];
  for (unsigned i = threadIdx.x; i < PB_LIN_BINS; i += GDN_BLOCK) s_h[i] = 0;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * GDN_BLOCK) {
    const unsigned c = cnt[i];
    if (c) atomicAdd(&s_h[c < PB_LIN_BINS - 1u ? c : PB_LIN_BINS - 1u], 1u);
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < PB_LIN_BINS; i += GDN_BLOCK)
    if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
}

// class of every source from its sampled count: 1 = hub (>= thr[0]), 1 + t = mid tier t (thr[t] <= count < thr[t-1]),
// 0 = main layout; the ids of every class are collected (unordered) in ids[class - 1]
struct PbTierArgs {
  unsigned thr[1 + PB_MAX_MID];
  uint32_t *ids[1 + PB_MAX_MID];
  unsigned cap[1 + PB_MAX_MID];
  int ntiers;  // classes in use (hub class included)
};
__global__ void __launch_bounds__(GDN_BLOCK)
pb_tier_class_kernel(const uint32_t *__restrict__ cnt, size_t n, PbTierArgs a, uint8_t *__restrict__ cls,
                     unsigned *__restrict__ n_ids /*1 + PB_MAX_MID*/) {
  const size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i >= n) return;
  const unsigned c = cnt[i];
  int k = 0;
  for (int t = a.ntiers - 1; t >= 0; t--)
    if (c >= a.thr[t]) k = t + 1;
  cls[i] = (uint8_t)k;
  if (k) {
    const unsigned pos = atomicAdd(&n_ids[k - 1], 1u);
    if (pos < a.cap[k - 1]) a.ids[k - 1][pos] = (uint32_t)i;
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
pb_count_rows_kernel(const eoff_t *__restrict__ rowptr, int32_t m, unsigned long long *__restrict__ out) {
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  const bool act = v < (unsigned)m && rowptr[v + 1] > rowptr[v];
  const unsigned long long mask = __ballot(act);
  if (gdn_lane() == 0 && mask) atomicAdd(out, (unsigned long long)__popcll(mask));
}

// picks the hub sources of `in_csr` (at most 2^15, each with >= PB_HUB_MIN_PER_BIN expected edges per bin) and, below
// them, up to max_mid MID tiers (each at most PB_MID_MAX sources, >= PB_MID_MIN_PER_BIN16 / 16 expected edges per bin).
// cls gets one byte per source id (0 main, 1 hub, 2.. mid tiers), hub_ids / mid_ids[t] the ascending ids.
#define PB_HUB_MIN_PER_BIN 2
#define PB_MID_MIN_PER_BIN16 4
// Host part of the tier choice: thresholds in count units of 2^sample_log edges (a count of the 1/16 row sample stands for 16)
// from the quarter-octave histogram h[PB_HUB_BUCKETS] and the linear histogram hl[PB_LIN_BINS] (nullable: no mid tiers)
// of the per-source counts.  thr[0] = hubs (0xFFFFFFFF: none qualifies; the slot is kept so that the classes keep
// their numbers), thr[1 + t] = mid tier t; *ntiers = classes in use, the hub class included.
void pb_choose_tiers(const unsigned *h, const unsigned *hl, uint64_t nbins, uint64_t nnz, int max_mid, unsigned min16,
                     unsigned *thr_out, int *ntiers_out, int sample_log = PB_HUB_SAMPLE_LOG) {
  struct {
    unsigned thr[1 + PB_MAX_MID];
    int ntiers;
  } ta;
  memset(&ta, 0, sizeof(ta));
  if (max_mid > PB_MAX_MID) max_mid = PB_MAX_MID;
  if (!hl) max_mid = 0;
  // a sampled count of c stands for about 16 c out-edges; a hub should have >= per_bin edges in an average bin
  uint64_t per_bin = PB_HUB_MIN_PER_BIN;
  if (const char *e = gdn_test_option("GDN_PB_HUB_MIN")) per_bin = (uint64_t)atoi(e) > 0 ? (uint64_t)atoi(e) : per_bin;  // tuning knob
  // smallest bucket >= the bucket of `want` whose sources, up to (not including) bucket `top`, number at most `cap`;
  // PB_HUB_BUCKETS = none
  auto pick = [&](uint64_t want, unsigned top, uint64_t cap) -> unsigned {
    if (want < 4) want = 4;
    if (want > 0x40000000ull) return PB_HUB_BUCKETS;
    unsigned bk = pb_hub_bucket((unsigned)want);
    if (pb_hub_bucket_floor(bk) < want) bk++;
    for (; bk < top; bk++) {
      uint64_t above = 0;
      for (unsigned j = bk; j < top; j++) above += h[j];
      if (above == 0) return PB_HUB_BUCKETS;
      if (above <= cap) return bk;
    }
    return PB_HUB_BUCKETS;
  };
  // tier 0 = hubs (the slot keeps an unreachable threshold when no source qualifies, so the classes keep their numbers)
  const unsigned bk0 = pick((nbins * per_bin) >> sample_log, PB_HUB_BUCKETS, 1u << PB_HUB_LOG);
  ta.thr[0] = bk0 < PB_HUB_BUCKETS ? pb_hub_bucket_floor(bk0) : 0xFFFFFFFFu;
  ta.ntiers = 1;
  uint64_t mid16 = min16 ? min16 : PB_MID_MIN_PER_BIN16;
  if (const char *e = gdn_xoption("GDN_PB_MID_MIN16")) mid16 = (uint64_t)atoi(e) > 0 ? (uint64_t)atoi(e) : mid16;  // tuning knob
  uint64_t mid_cap = PB_MID_MAX;  // GDN_PB_MID_CAP: test knob (fewer sources per tier, so that small graphs get two tiers)
  if (const char *e = gdn_test_option("GDN_PB_MID_CAP")) mid_cap = (uint64_t)atoi(e) > 0 && (uint64_t)atoi(e) < PB_MID_MAX ? (uint64_t)atoi(e) : mid_cap;
  if (max_mid > 0) {
    // mid tiers: consecutive count ranges [thr[t], thr[t-1]) below the hubs, each filled up to mid_cap sources, down
    // to the count that stands for mid16 / 16 edges per average bin
    uint64_t n_hub_src = 0;
    for (unsigned j = bk0; j < PB_HUB_BUCKETS; j++) n_hub_src += h[j];
    // sources with a count in [c, top): the last linear bin also holds everything beyond it, the hubs included
    uint64_t top = bk0 < PB_HUB_BUCKETS ? ta.thr[0] : 0xFFFFFFFFull;
    uint64_t want = (nbins * mid16) >> (sample_log + 4);
    if (want < (64u >> sample_log)) want = 64u >> sample_log;  // 64 edges (4 sampled): below, a table line is fetched per record
    for (int t = 0; t < max_mid && top > want; t++) {
      uint64_t acc = 0, thr = top;
      for (uint64_t c = (top < PB_LIN_BINS ? top : PB_LIN_BINS) - 1;; c--) {
        uint64_t here = hl[c];
        if (c == PB_LIN_BINS - 1) {  // the open-ended bin: not cut inside; take it whole or not at all
          if (top <= c) here = 0;
          else here -= (here >= n_hub_src ? n_hub_src : here);
        }
        if (acc + here > mid_cap) break;
        acc += here;
        thr = c;
        if (c <= want) break;
      }
      if (thr >= top || acc == 0) break;
      // a tier of a few thousand sources (a graph without skew: the tail of a Poisson degree distribution) costs every
      // bin a stream and a table for a fraction of a percent of the edges: a tier has to stand for >= 1/64 of them
      {
        unsigned long long est = 0;  // a sampled count of c ~ 16 c out-edges
        for (uint64_t c = thr; c < (top < PB_LIN_BINS ? top : PB_LIN_BINS); c++) est += ((unsigned long long)c << sample_log) * hl[c];
        if (est * 64ull < nnz) break;
      }
      ta.thr[1 + t] = (unsigned)thr;
      ta.ntiers = 2 + t;
      top = thr;
    }
  }
  for (int t = 0; t < 1 + PB_MAX_MID; t++) thr_out[t] = ta.thr[t];
  *ntiers_out = ta.ntiers;
}

int pb_pick_tiers(const gdn_graph *g, int32_t m_global, int log_bin, DevBuf<uint8_t> &cls, DevBuf<uint32_t> &hub_ids,
                  unsigned *n_hubs, int max_mid, DevBuf<uint32_t> *mid_ids, unsigned *n_mid, unsigned min16) {
  *n_hubs = 0;
  if (max_mid > PB_MAX_MID) max_mid = PB_MAX_MID;
  for (int t = 0; t < max_mid; t++) n_mid[t] = 0;
  DevBuf<uint32_t> cnt;
  DevBuf<unsigned> hist;
  DevBuf<unsigned long long> nrows;
  GDN_TRY(cnt.alloc((size_t)m_global));
  GDN_TRY(hist.alloc(PB_HUB_BUCKETS + 1 + PB_MAX_MID));
  GDN_TRY(nrows.alloc(1));
  GDN_HIP(hipMemset(cnt.p, 0, (size_t)m_global * 4));
  GDN_HIP(hipMemset(hist.p, 0, (PB_HUB_BUCKETS + 1 + PB_MAX_MID) * 4));
  GDN_HIP(hipMemset(nrows.p, 0, 8));
  const uint64_t sampled = ((uint64_t)g->m + (1u << PB_HUB_SAMPLE_LOG) - 1) >> PB_HUB_SAMPLE_LOG;
  hipLaunchKernelGGL(pb_hub_sample_kernel, dim3(gdn_nblocks(sampled * 64)), dim3(GDN_BLOCK), 0, 0, g->rowptr, g->colidx,
                     g->m, cnt.p);
  hipLaunchKernelGGL(pb_hub_hist_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, cnt.p, (size_t)m_global, hist.p);
  hipLaunchKernelGGL(pb_count_rows_kernel, dim3(gdn_nblocks((uint64_t)g->m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, g->m,
                     nrows.p);
  GDN_HIP(hipGetLastError());
  unsigned h[PB_HUB_BUCKETS];
  unsigned long long active_rows = 0;
  GDN_HIP(hipMemcpy(h, hist.p, sizeof(h), hipMemcpyDeviceToHost));
  GDN_HIP(hipMemcpy(&active_rows, nrows.p, 8, hipMemcpyDeviceToHost));
  const uint64_t nbins = ((active_rows + (1ull << log_bin) - 1) >> log_bin) + 1;
  PbTierArgs ta;
  memset(&ta, 0, sizeof(ta));
  std::vector<unsigned> hl;
  if (max_mid > 0) {
    DevBuf<unsigned> lin;
    GDN_TRY(lin.alloc(PB_LIN_BINS));
    GDN_HIP(hipMemset(lin.p, 0, PB_LIN_BINS * 4));
    hipLaunchKernelGGL(pb_hub_hist_lin_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, cnt.p, (size_t)m_global, lin.p);
    GDN_HIP(hipGetLastError());
    hl.resize(PB_LIN_BINS);
    GDN_HIP(hipMemcpy(hl.data(), lin.p, PB_LIN_BINS * 4, hipMemcpyDeviceToHost));
  }
  pb_choose_tiers(h, max_mid > 0 ? hl.data() : nullptr, nbins, g->nnz, max_mid, min16, ta.thr, &ta.ntiers);
  ta.cap[0] = 1u << PB_HUB_LOG;
  for (int t = 1; t < ta.ntiers; t++) ta.cap[t] = PB_MID_MAX;
  const bool no_hubs = ta.thr[0] == 0xFFFFFFFFu;
  if (no_hubs && ta.ntiers == 1) return GDN_OK;
  GDN_TRY(cls.alloc((size_t)m_global));
  GDN_TRY(hub_ids.alloc(1u << PB_HUB_LOG));
  ta.ids[0] = hub_ids.p;
  for (int t = 1; t < ta.ntiers; t++) {
    GDN_TRY(mid_ids[t - 1].alloc(PB_MID_MAX));
    ta.ids[t] = mid_ids[t - 1].p;
  }
  hipLaunchKernelGGL(pb_tier_class_kernel, dim3(gdn_nblocks((uint64_t)m_global)), dim3(GDN_BLOCK), 0, 0, cnt.p,
                     (size_t)m_global, ta, cls.p, hist.p + PB_HUB_BUCKETS);
  GDN_HIP(hipGetLastError());
  unsigned n[1 + PB_MAX_MID];
  GDN_HIP(hipMemcpy(n, hist.p + PB_HUB_BUCKETS, sizeof(n), hipMemcpyDeviceToHost));
  std::vector<uint32_t> ids;
  for (int t = 0; t < ta.ntiers; t++) {
    if (n[t] == 0) continue;
    if (n[t] > ta.cap[t]) {  // cannot happen: the histogram counted them
      gdn_set_error("pb_pick_tiers: tier %d holds %u sources (cap %u)", t, n[t], ta.cap[t]);
      return GDN_ERR_INVALID;
    }
    // source k of a tier = the k-th source of its class in id order = its compact index in the tier's layout
    ids.resize(n[t]);
    GDN_HIP(hipMemcpy(ids.data(), ta.ids[t], (size_t)n[t] * 4, hipMemcpyDeviceToHost));
    std::sort(ids.begin(), ids.end());
    GDN_HIP(hipMemcpy(ta.ids[t], ids.data(), (size_t)n[t] * 4, hipMemcpyHostToDevice));
    if (t == 0) *n_hubs = n[0];
    else n_mid[t - 1] = n[t];
  }
  if (gdn_xoption("GDN_PB_TRACE")) {
    fprintf(stderr, "[pb_pick_tiers] bins %llu: hubs %u (sampled count >= %u)", (unsigned long long)nbins, n[0], ta.thr[0]);
    for (int t = 1; t < ta.ntiers; t++) fprintf(stderr, ", mid %d: %u (>= %u)", t, n[t], ta.thr[t]);
    fprintf(stderr, "\n");
  }
  return GDN_OK;
}

int pb_pick_hubs(const gdn_graph *g, int32_t m_global, int log_bin, DevBuf<uint8_t> &cls, DevBuf<uint32_t> &hub_ids,
                 unsigned *n_hubs) {
  return pb_pick_tiers(g, m_global, log_bin, cls, hub_ids, n_hubs, 0, nullptr, nullptr);
}

// ---- mid tiers: the layout pb_build made for the sources of one mid class (chunks of 2^15 sources, tiles sorted by
// (source, row)) becomes ONE bin-major stream of 32-bit records (source index in the tier << 14 | row in the bin);
// pad records point at the tier's zero slot.  U, V and G are not needed afterwards.
__global__ void __launch_bounds__(GDN_BLOCK)
pb_mid_records_kernel(const uint16_t *__restrict__ U, const uint16_t *__restrict__ V, const uint32_t *__restrict__ G,
                      const eoff_t *__restrict__ chunk_ptr, unsigned nchunks, unsigned long long ngroups, int log_group,
                      int log_chunk, unsigned pad_id, unsigned zslot, uint32_t *__restrict__ rec,
                      const float *__restrict__ ev_in, float *__restrict__ ev_out) {
  const unsigned long long g = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (g >= ngroups) return;
  const uint32_t dst = G[g];
  if ((unsigned long long)dst >= ngroups) return;  // a group of an alignment gap (dump group)
  const eoff_t e0 = (eoff_t)g << log_group;
  unsigned c = 0;
  while (c + 1 < nchunks && chunk_ptr[c + 1] <= e0) c++;
  const unsigned grp = 1u << log_group;
  for (unsigned i = 0; i < grp; i++) {
    const unsigned u = U[e0 + i];
    const unsigned k = (u == pad_id) ? zslot : ((c << log_chunk) + u);
    rec[((size_t)dst << log_group) + i] = (k << PB_MID_ROW_BITS) | (unsigned)V[((size_t)dst << log_group) + i];
    if (ev_in) ev_out[((size_t)dst << log_group) + i] = ev_in[e0 + i];
  }
}

int pb_mid_finish(PbPlan &p, unsigned n_src, DevBuf<uint32_t> &rec, DevBuf<float> *ev) {
  GDN_REQUIRE(p.log_bin <= PB_MID_ROW_BITS && n_src <= PB_MID_MAX, "mid tier: row / source index width");
  GDN_REQUIRE(((size_t)p.nchunks << p.log_chunk) >= n_src && p.chunk_slots == (1u << p.log_chunk), "mid tier: chunks");
  const unsigned grp = 1u << p.log_group;
  GDN_TRY(rec.alloc(p.n_pad + grp));
  const unsigned long long ngroups = p.n_pad >> p.log_group;
  const unsigned zrec = n_src << PB_MID_ROW_BITS;
  const unsigned long long fb = (p.n_pad + grp + GDN_BLOCK - 1) / GDN_BLOCK;
  hipLaunchKernelGGL(pb_fill_u32_kernel, dim3((unsigned)(fb > 262144ull ? 262144ull : fb)), dim3(GDN_BLOCK), 0, 0, rec.p,
                     p.n_pad + grp, zrec);
  DevBuf<float> ev_b;  // the per-edge values in record (bin-major) order
  if (ev) {
    GDN_TRY(ev_b.alloc(p.n_pad + grp));
    GDN_HIP(hipMemset(ev_b.p, 0, (p.n_pad + grp) * sizeof(float)));
  }
  if (ngroups)
    hipLaunchKernelGGL(pb_mid_records_kernel, dim3(gdn_nblocks(ngroups)), dim3(GDN_BLOCK), 0, 0, p.U.p, p.V.p, p.G.p,
                       p.chunk_ptr.p, p.nchunks, ngroups, p.log_group, p.log_chunk, p.chunk_slots, n_src, rec.p,
                       ev ? ev->p : nullptr, ev ? ev_b.p : nullptr);
  GDN_HIP(hipGetLastError());
  GDN_HIP(hipDeviceSynchronize());
  if (ev) {
    ev->take(ev_b);
  }
  p.U.release();
  p.V.release();
  p.G.release();
  return GDN_OK;
}


// ---- hub ROWS (PageRank): the rows with the most in-edges, at most max_rows of them and each with >= min_deg
// in-edges; dcls gets one byte per row, row_ids the ascending ids.  Degrees are exact (row offsets).
__global__ void __launch_bounds__(GDN_BLOCK)
pb_hubrow_hist_kernel(const eoff_t *__restrict__ rowptr, int32_t m, unsigned *__restrict__ hist) {
  __shared__ unsigned s_h[PB_HUB_BUCKETS];
  if (threadIdx.x < PB_HUB_BUCKETS) s_h[threadIdx.x] = 0;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < (size_t)m; i += (size_t)gridDim.x * GDN_BLOCK) {
    const eoff_t d = rowptr[i + 1] - rowptr[i];
    if (d) atomicAdd(&s_h[pb_hub_bucket(d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)d)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < PB_HUB_BUCKETS && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}

__global__ void __launch_bounds__(GDN_BLOCK)
pb_hubrow_class_kernel(const eoff_t *__restrict__ rowptr, int32_t m, unsigned thr, uint8_t *__restrict__ dcls,
                       uint32_t *__restrict__ ids, unsigned cap, unsigned *__restrict__ n_ids) {
  const size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i >= (size_t)m) return;
  const bool hub = rowptr[i + 1] - rowptr[i] >= (eoff_t)thr;
  dcls[i] = hub ? 1 : 0;
  if (hub) {
    const unsigned pos = atomicAdd(n_ids, 1u);
    if (pos < cap) ids[pos] = (uint32_t)i;
  }
}

int pb_pick_hub_rows(const gdn_graph *g, unsigned max_rows, uint64_t min_deg, DevBuf<uint8_t> &dcls, DevBuf<uint32_t> &row_ids,
                     unsigned *n_rows) {
  *n_rows = 0;
  if (max_rows == 0) return GDN_OK;
  DevBuf<unsigned> hist;
  GDN_TRY(hist.alloc(PB_HUB_BUCKETS + 1));
  GDN_HIP(hipMemset(hist.p, 0, (PB_HUB_BUCKETS + 1) * 4));
  hipLaunchKernelGGL(pb_hubrow_hist_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, g->rowptr, g->m, hist.p);
  GDN_HIP(hipGetLastError());
  unsigned h[PB_HUB_BUCKETS];
  GDN_HIP(hipMemcpy(h, hist.p, sizeof(h), hipMemcpyDeviceToHost));
  if (min_deg < 4) min_deg = 4;
  if (min_deg > 0x40000000ull) return GDN_OK;
  unsigned bk = pb_hub_bucket((unsigned)min_deg);
  if (pb_hub_bucket_floor(bk) < min_deg) bk++;
  for (;; bk++) {
    if (bk >= PB_HUB_BUCKETS) return GDN_OK;
    uint64_t above = 0;
    for (unsigned j = bk; j < PB_HUB_BUCKETS; j++) above += h[j];
    if (above == 0) return GDN_OK;
    if (above <= max_rows) break;
  }
  GDN_TRY(dcls.alloc((size_t)g->m));
  GDN_TRY(row_ids.alloc(max_rows));
  hipLaunchKernelGGL(pb_hubrow_class_kernel, dim3(gdn_nblocks((uint64_t)g->m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, g->m,
                     pb_hub_bucket_floor(bk), dcls.p, row_ids.p, max_rows, hist.p + PB_HUB_BUCKETS);
  GDN_HIP(hipGetLastError());
  unsigned n = 0;
  GDN_HIP(hipMemcpy(&n, hist.p + PB_HUB_BUCKETS, 4, hipMemcpyDeviceToHost));
  if (n == 0 || n > max_rows) return GDN_OK;
  std::vector<uint32_t> ids(n);
  GDN_HIP(hipMemcpy(ids.data(), row_ids.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  std::sort(ids.begin(), ids.end());  // hub row k = k-th marked row in id order = its compact row index in the hub-row layout
  GDN_HIP(hipMemcpy(row_ids.p, ids.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  *n_rows = n;
  return GDN_OK;
}

int pb_order_bins_by_work(PbPlan &main, int n_tiers, const eoff_t *const *tier_bin_ptr, double main_bytes_per_edge,
                          double rec_bytes, double row_bytes) {
  const unsigned nb = main.nbins;
  std::vector<eoff_t> bp((size_t)nb + 1);
  std::vector<uint32_t> blo((size_t)nb + 1, 0u);
  std::vector<double> work(nb, 0.0);
  GDN_HIP(hipMemcpy(bp.data(), main.bin_ptr.p, bp.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
  for (unsigned b = 0; b < nb; b++) work[b] = main_bytes_per_edge * (double)(bp[b + 1] - bp[b]);
  if (main.compact) {
    GDN_HIP(hipMemcpy(blo.data(), main.bin_lo.p, blo.size() * 4, hipMemcpyDeviceToHost));
    for (unsigned b = 0; b < nb; b++) work[b] += row_bytes * (double)(blo[b + 1] - blo[b]);
  }
  for (int t = 0; t < n_tiers; t++) {
    GDN_HIP(hipMemcpy(bp.data(), tier_bin_ptr[t], bp.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
    for (unsigned b = 0; b < nb; b++) work[b] += rec_bytes * (double)(bp[b + 1] - bp[b]);
  }
  std::vector<uint32_t> bo(nb);
  for (unsigned b = 0; b < nb; b++) bo[b] = b;
  std::stable_sort(bo.begin(), bo.end(), [&](uint32_t a, uint32_t b) { return work[a] > work[b]; });
  GDN_HIP(hipMemcpy(main.bin_order.p, bo.data(), (size_t)nb * 4, hipMemcpyHostToDevice));
  if (gdn_xoption("GDN_PB_TRACE") && nb) {
    double mx = 0, sum = 0;
    for (unsigned b = 0; b < nb; b++) {
      sum += work[b];
      if (work[b] > mx) mx = work[b];
    }
    fprintf(stderr, "[pb_order_bins] bins %u: %.1f MB each on average, largest %.1f MB\n", nb, sum / nb / 1e6, mx / 1e6);
  }
  return GDN_OK;
}

// slots used per slice when n_act active vertices are spread over whole rounds of workgroups (see pb_build)
uint64_t pb_slots_per_slice(uint64_t n_act, int lg, int lg_full) {
  int ncu = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (ncu <= 0) ncu = 256;
  const char *be = gdn_xoption("GDN_PB_BALANCE");  // 0 switches the spreading off (A/B measurements)
  const bool on = !(be && be[0] == '0');
  const uint64_t cap = 1ull << lg;
  const uint64_t ns = (n_act + cap - 1) >> lg;
  // fewer slices than CUs (and at least 1024 slots each): exactly ONE round of smaller slices (round 6; GDN_PB_FILL_ROUND=0 in
  // an experiments build keeps the cap)
  const char *fe = gdn_xoption("GDN_PB_FILL_ROUND");
  if (on && lg == lg_full && ns < (uint64_t)ncu && n_act >= (uint64_t)ncu * 1024u && !(fe && fe[0] == '0')) {
    uint64_t per1 = (n_act + (uint64_t)ncu - 1) / (uint64_t)ncu;
    per1 = (per1 + 3) & ~3ull;
    return per1 < cap ? per1 : cap;
  }
  if (!on || lg != lg_full || ns <= (uint64_t)ncu) return cap;
  const uint64_t rounds = (ns + (uint64_t)ncu - 1) / (uint64_t)ncu;
  uint64_t per = (n_act + rounds * ncu - 1) / (rounds * ncu);
  per = (per + 3) & ~3ull;
  return per < cap ? per : cap;
}

int pb_build(const gdn_graph *g, int32_t m_global, int log_chunk, int log_bin, PbPlan &p, bool alloc_vals,
             const float *edge_vals_in, DevBuf<float> *edge_vals_out, bool compact, bool rows_are_sources, unsigned pad,
             int log_group, const uint8_t *src_class, int want_class, bool src_major, bool v_delta,
             const uint8_t *dst_class, int want_dst, bool rows_of_class_only, bool no_gaps, int bin_balance_log,
             PbScratch *scratch) {
  GDN_REQUIRE(log_chunk >= 8 && log_chunk <= 15, "log_chunk");
  GDN_REQUIRE(!v_delta || (pad >= 32 && !src_major && !rows_are_sources), "delta-coded rows: tiles of whole 32-edge groups, sorted by row");
  // (source classes on an out-CSR -- SSSP's sweeps -- index the CSR's ROWS: the key visitor swaps row and column first;
  // what an out-CSR cannot have is the compaction, see below)
  GDN_REQUIRE(log_group >= 3 && log_group <= 7 && pad >= (1u << log_group) && pad <= 128 && (pad & (pad - 1)) == 0,
              "pad / log_group");
  const auto t_begin = std::chrono::steady_clock::now();
  p.log_group = log_group;
  const unsigned grp = 1u << log_group;  // u16 local ids + one pad value
  GDN_REQUIRE(log_bin >= 8 && log_bin <= 15, "log_bin");
  GDN_REQUIRE(!(compact && rows_are_sources), "compaction is not supported on an out-CSR");
  const int32_t m = g->m;
  const unsigned long long n = g->nnz;
  unsigned long long n_use = n;  // edges that belong to this layout (< n with a source-class filter)
  // normal: rows = destinations (m), columns = sources (m_global).  out-CSR: rows = sources (m),
  // columns = destinations (m_global)
  p.m_local = rows_are_sources ? m_global : m;
  p.m_global = rows_are_sources ? m : m_global;
  p.nnz = n;
  p.log_chunk = log_chunk;
  p.log_bin = log_bin;
  p.compact = compact;
  p.chunk_slots = 1u << log_chunk;
  uint64_t n_src = (uint64_t)p.m_global, n_dst = (uint64_t)p.m_local;
  DevBuf<eoff_t> cs, cd;  // compact index of every source id / row (exclusive scans of the flags)
  DevBuf<uint32_t> inv_s, inv_d;  // compact index -> original id (only built to look edge values up again)
  if (compact) {
    DevBuf<uint32_t> sflag, dflag;
    DevBuf<unsigned long long> bigitems;
    DevBuf<unsigned> cnt;
    const uint64_t bigcap64 = n / EXP_CHUNK + (uint64_t)m / 64 + 1024;
    const unsigned bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
    GDN_TRY(sflag.alloc((size_t)m_global));
    GDN_TRY(dflag.alloc((size_t)m));
    GDN_TRY(cs.alloc((size_t)m_global + 1));
    GDN_TRY(cd.alloc((size_t)m + 1));
    GDN_TRY(bigitems.alloc(bigcap));
    GDN_TRY(cnt.alloc(2));
    GDN_HIP(hipMemset(cnt.p, 0, 8));
    GDN_HIP(hipMemset(sflag.p, 0, (size_t)m_global * 4));
    ExpBigList big;
    big.items = bigitems.p;
    big.capacity = bigcap;
    big.count = cnt.p;
    big.overflow = cnt.p + 1;
    PbMarkVis mv;
    mv.colidx = g->colidx;
    mv.mark = sflag.p;
    mv.cls = src_class;
    mv.want = want_class;
    hipLaunchKernelGGL(pb_mark_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, big, mv, dflag.p,
                       dst_class, want_dst, (dst_class && rows_of_class_only) ? 1 : 0);
    hipLaunchKernelGGL(pb_mark_big_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, mv);
    GDN_HIP(hipGetLastError());
    GDN_TRY(gdn_exclusive_scan_u32_to_u64(sflag.p, cs.p, (size_t)m_global, 0));
    GDN_TRY(gdn_exclusive_scan_u32_to_u64(dflag.p, cd.p, (size_t)m, 0));
    eoff_t tot[2];
    GDN_HIP(hipMemcpy(&tot[0], cs.p + m_global, sizeof(eoff_t), hipMemcpyDeviceToHost));
    GDN_HIP(hipMemcpy(&tot[1], cd.p + m, sizeof(eoff_t), hipMemcpyDeviceToHost));
    unsigned ovf[2];
    GDN_HIP(hipMemcpy(ovf, cnt.p, 8, hipMemcpyDeviceToHost));
    if (ovf[1]) {
      gdn_set_error("pb_build: device worklist overflow");
      return GDN_ERR_OVERFLOW;
    }
    n_src = tot[0];
    n_dst = tot[1];
    {
      // Whole rounds: a full-size slice (128 KB of LDS) means one workgroup per CU, so a phase takes
      // ceil(slices / CUs) rounds of about equal length -- 1585 chunks on 256 CUs are 6.2 rounds of work in the
      // time of 7.  Spread the active vertices over rounds * CUs slices instead (fewer than 2^log slots used per
      // slice; the compact index keeps its power-of-two slice stride so every shift below stays valid).
      const uint64_t per_c = pb_slots_per_slice(n_src, log_chunk, PB_MAX_LOG_CHUNK);
      const uint64_t per_b = pb_slots_per_slice(n_dst, log_bin, bin_balance_log);
      p.chunk_slots = (unsigned)per_c;
      if (per_c < (1ull << log_chunk)) {
        hipLaunchKernelGGL(pb_respace_kernel, dim3(gdn_nblocks((uint64_t)m_global + 1)), dim3(GDN_BLOCK), 0, 0, cs.p,
                           (size_t)m_global + 1, per_c, log_chunk);
        n_src = ((n_src + per_c - 1) / per_c) << log_chunk;
      }
      if (per_b < (1ull << log_bin)) {
        hipLaunchKernelGGL(pb_respace_kernel, dim3(gdn_nblocks((uint64_t)m + 1)), dim3(GDN_BLOCK), 0, 0, cd.p, (size_t)m + 1,
                           per_b, log_bin);
        n_dst = ((n_dst + per_b - 1) / per_b) << log_bin;
      }
      GDN_HIP(hipGetLastError());
    }
    const unsigned nch = (unsigned)((n_src + (1u << log_chunk) - 1) >> log_chunk), nbn = (unsigned)((n_dst + (1u << log_bin) - 1) >> log_bin);
    const unsigned nchunks = nch ? nch : 1u, nbins = nbn ? nbn : 1u;
    GDN_TRY(p.src_bits.alloc(((size_t)m_global + 31) / 32 + 1));
    GDN_TRY(p.dst_bits.alloc(((size_t)m + 31) / 32 + 1));
    GDN_TRY(p.chunk_lo.alloc((size_t)nchunks + 1));
    GDN_TRY(p.bin_lo.alloc((size_t)nbins + 1));
    GDN_HIP(hipMemset(p.chunk_lo.p, 0, ((size_t)nchunks + 1) * 4));
    GDN_HIP(hipMemset(p.bin_lo.p, 0, ((size_t)nbins + 1) * 4));
    hipLaunchKernelGGL(pb_slices_kernel, dim3(gdn_nblocks(((uint64_t)m_global + 31) / 32 + 1)), dim3(GDN_BLOCK), 0, 0,
                       sflag.p, cs.p, (unsigned)m_global, log_chunk, nchunks, p.src_bits.p, p.chunk_lo.p);
    hipLaunchKernelGGL(pb_slices_kernel, dim3(gdn_nblocks(((uint64_t)m + 31) / 32 + 1)), dim3(GDN_BLOCK), 0, 0, dflag.p,
                       cd.p, (unsigned)m, log_bin, nbins, p.dst_bits.p, p.bin_lo.p);
    if (edge_vals_in) {
      GDN_TRY(inv_s.alloc((size_t)nchunks << log_chunk));
      GDN_TRY(inv_d.alloc((size_t)nbins << log_bin));
      hipLaunchKernelGGL(pb_inverse_kernel, dim3(gdn_nblocks((uint64_t)m_global)), dim3(GDN_BLOCK), 0, 0, sflag.p, cs.p,
                         (size_t)m_global, inv_s.p);
      hipLaunchKernelGGL(pb_inverse_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, dflag.p, cd.p, (size_t)m,
                         inv_d.p);
    }
    GDN_HIP(hipGetLastError());
    GDN_HIP(hipDeviceSynchronize());
  }
  p.nchunks = (unsigned)((n_src + (1u << log_chunk) - 1) >> log_chunk);
  p.nbins = (unsigned)((n_dst + (1u << log_bin) - 1) >> log_bin);
  if (p.nchunks == 0) p.nchunks = 1;
  if (p.nbins == 0) p.nbins = 1;
  const int bin_bits = bits_for((int32_t)p.nbins);
  const int chunk_bits = bits_for((int32_t)p.nchunks);
  const unsigned long long ntiles = (unsigned long long)p.nchunks * p.nbins;
  const unsigned long long gb = (n + GDN_BLOCK - 1) / GDN_BLOCK;
  const unsigned grid_n = (unsigned)(gb > 262144ull ? 262144ull : (gb ? gb : 1));
  // The long-lived streamed arrays are allocated FIRST, at an upper bound of the padded size, before
  // the multi-GB sort temporaries come and go: allocated afterwards they land in whatever fragments
  // the temporaries left behind (experiment GDN_PB_EARLY_ALLOC)
  const uint64_t n_pad_bound = ((n + 15) & ~15ull) + 16ull * (ntiles < n ? ntiles : n) + 64;
  const bool early = false;  // measured: no effect (7.18 vs 7.23 ms), kept off
  if (early) {
    GDN_TRY(p.U.alloc(n_pad_bound));
    GDN_TRY(p.V.alloc(n_pad_bound));
    GDN_TRY(p.G.alloc((n_pad_bound >> log_group) + 1));
    if (alloc_vals) GDN_TRY(p.vals.alloc(n_pad_bound));
  }
  DevBuf<eoff_t> tsu, pu, pv;
  DevBuf<uint32_t> psz_c, psz_b;
  GDN_TRY(tsu.alloc(ntiles + 1));
  GDN_TRY(pu.alloc(ntiles + 1));
  GDN_TRY(pv.alloc(ntiles + 1));
  GDN_TRY(psz_c.alloc(ntiles));
  GDN_TRY(psz_b.alloc(ntiles));
  GDN_TRY(p.chunk_ptr.alloc((size_t)p.nchunks + 1));
  GDN_TRY(p.bin_ptr.alloc((size_t)p.nbins + 1));
  GDN_TRY(p.errflag.alloc(1));
  GDN_HIP(hipMemset(p.errflag.p, 0, sizeof(unsigned)));
  {
    DevBuf<unsigned long long> ka_own, kb_own, bigitems, nvalid;
    DevBuf<unsigned long long> &ka = scratch ? scratch->ka : ka_own, &kb = scratch ? scratch->kb : kb_own;
    DevBuf<unsigned> cnt;
    const uint64_t bigcap64 = n / EXP_CHUNK + (uint64_t)m / 64 + 1024;
    const unsigned bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
    if (ka.n < n || !ka.p) GDN_TRY(ka.alloc_scratch(n));
    if (kb.n < n || !kb.p) GDN_TRY(kb.alloc_scratch(n));
    GDN_TRY(bigitems.alloc(bigcap));
    GDN_TRY(cnt.alloc(2));
    GDN_TRY(nvalid.alloc(1));
    GDN_HIP(hipMemset(cnt.p, 0, 8));
    GDN_HIP(hipMemset(nvalid.p, 0, 8));
    ExpBigList big;
    big.items = bigitems.p;
    big.capacity = bigcap;
    big.count = cnt.p;
    big.overflow = cnt.p + 1;
    const unsigned key_bits = (unsigned)(chunk_bits + bin_bits + log_chunk + log_bin);
    PbKeyVis vis;
    vis.colidx = g->colidx;
    vis.keys = ka.p;
    vis.cs = compact ? cs.p : nullptr;
    vis.cd = compact ? cd.p : nullptr;
    vis.log_chunk = log_chunk;
    vis.log_bin = log_bin;
    vis.bin_bits = bin_bits;
    vis.transposed = rows_are_sources ? 1 : 0;
    vis.v = 0;
    vis.cls = src_class;
    vis.want = want_class;
    vis.dcls = dst_class;
    vis.dwant = want_dst;
    vis.src_major = src_major ? 1 : 0;
    vis.sentinel = 1ull << key_bits;
    vis.nvalid_out = (src_class || dst_class) ? nvalid.p : nullptr;
    hipLaunchKernelGGL(pb_keys_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, big, vis);
    hipLaunchKernelGGL(pb_keys_big_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
    GDN_HIP(hipGetLastError());
    unsigned h[2];
    GDN_HIP(hipMemcpy(h, cnt.p, 8, hipMemcpyDeviceToHost));
    if (h[1]) {
      gdn_set_error("pb_build: device worklist overflow");
      return GDN_ERR_OVERFLOW;
    }
    bigitems.release();
    cs.release();
    cd.release();
    if (src_class || dst_class) {  // the edges of the other classes carry the sentinel key: they sort behind the n_use real keys
      GDN_HIP(hipMemcpy(&n_use, nvalid.p, 8, hipMemcpyDeviceToHost));
      p.nnz = n_use;
    }
    const unsigned long long *sorted = nullptr;
    // The keys were written in CSR order = (bin, row, chunk, source) ascending (rows ascending, columns ascending in a
    // row; the compact indices are monotone): the tile order (chunk, bin, row, source) is a STABLE sort by the chunk field
    // alone -- 11-12 bits instead of ~58, two radix passes instead of eight.  (A caller's row with unsorted columns only
    // changes the order of one row's edges inside a tile, which nothing depends on.)  Not for the out-CSR form and the
    // (source, row) tile order, whose inner order CSR order does not give.
    const bool by_chunk_only = !rows_are_sources && !src_major && !gdn_xoption("GDN_PB_FULL_SORT");
    const bool filtered = src_class || dst_class;
    if (!by_chunk_only && filtered && n_use * 2 < n && (ka.p && kb.p)) {
      // a class-filtered layout sorted on its whole key (the tiers: 9-24 % of the edges each): ONE pass on the sentinel
      // bit moves the real keys to the front (stable partition), the other passes sort those only
      const unsigned long long *part = nullptr;
      GDN_TRY(gdn_radix_sort_u64(ka.p, kb.p, n, key_bits, key_bits + 1u, &part));
      unsigned long long *front = const_cast<unsigned long long *>(part), *other = part == ka.p ? kb.p : ka.p;
      GDN_TRY(gdn_radix_sort_u64(front, other, n_use, 0u, key_bits, &sorted));
      if (scratch == nullptr) {  // free the buffer that does not hold the result
        if (sorted == ka.p) kb.release();
        else ka.release();
      }
    } else {
      GDN_TRY(sort_keys(ka, kb, n, key_bits + (filtered ? 1u : 0u), &sorted, /*keep_both=*/scratch != nullptr,
                        by_chunk_only ? key_bits - (unsigned)chunk_bits : 0u));
    }
    if (n_use == 0) {
      hipLaunchKernelGGL(pb_fill_u64_kernel, dim3(gdn_nblocks(ntiles + 1)), dim3(GDN_BLOCK), 0, 0, tsu.p, ntiles + 1,
                         (eoff_t)0);
    } else {
      hipLaunchKernelGGL(pb_bounds_kernel, dim3(grid_n), dim3(GDN_BLOCK), 0, 0, sorted, n_use, log_chunk + log_bin, bin_bits,
                         p.nbins, ntiles, tsu.p);
    }
    // tile runs are padded to 16 edges = whole 64-byte lines of vals, so phase A never leaves a
    // partially written line to another workgroup (measured 7.9 vs 8.4 ms/iter at pad 8, RMAT-27)
    DevBuf<uint32_t> nd;  // delta-coded rows: fillers in front of every sorted key, and their exclusive scan
    DevBuf<eoff_t> ds;
    if (v_delta) {
      GDN_TRY(nd.alloc(n_use + 1));
      GDN_TRY(ds.alloc(n_use + 2));
      if (n_use)
        hipLaunchKernelGGL(pb_gap_kernel, dim3(grid_n), dim3(GDN_BLOCK), 0, 0, sorted, n_use, log_chunk, log_bin, nd.p);
      GDN_HIP(hipGetLastError());
      GDN_TRY(gdn_exclusive_scan_u32_to_u64(nd.p, ds.p, (size_t)n_use, 0));
    }
    hipLaunchKernelGGL(pb_tile_sizes_kernel, dim3(gdn_nblocks(ntiles)), dim3(GDN_BLOCK), 0, 0, tsu.p, p.nchunks, p.nbins,
                       pad, psz_c.p, psz_b.p, v_delta ? ds.p : nullptr);
    GDN_HIP(hipGetLastError());
    GDN_TRY(gdn_exclusive_scan_u32_to_u64(psz_c.p, pu.p, (size_t)ntiles, 0));
    GDN_TRY(gdn_exclusive_scan_u32_to_u64(psz_b.p, pv.p, (size_t)ntiles, 0));
    hipLaunchKernelGGL(pb_ptrs_kernel, dim3(gdn_nblocks((uint64_t)(p.nchunks > p.nbins ? p.nchunks : p.nbins) + 1)),
                       dim3(GDN_BLOCK), 0, 0, pu.p, pv.p, p.nchunks, p.nbins, p.chunk_ptr.p, p.bin_ptr.p);
    GDN_HIP(hipGetLastError());
    // Every chunk's range in U/G and every bin's range in V/vals starts on an aligned boundary: streams that
    // begin at arbitrary offsets run 10-20 % slower on this memory system (measured: phase A 3.70 -> 2.98 ms,
    // phase B stream 2.26 -> 2.05 ms on RMAT-27 with 16384-edge alignment, profiles/pb_ablation_r01.txt).
    // The gap behind a slice belongs to the slice: U = pad id, G = the dump group, V = 0, scratch = neutral.
    eoff_t n_pad = 0;
    {
      std::vector<eoff_t> cs((size_t)p.nchunks + 1), bs((size_t)p.nbins + 1), du(p.nchunks), dv(p.nbins);
      GDN_HIP(hipMemcpy(cs.data(), p.chunk_ptr.p, cs.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
      GDN_HIP(hipMemcpy(bs.data(), p.bin_ptr.p, bs.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
      auto pick_align = [](eoff_t total, unsigned parts) {
        eoff_t a = 16;
        while (a < 16384 && a * 32 <= total / (parts ? parts : 1)) a <<= 1;  // keep the gaps below ~3 % of a slice
        return a;
      };
      // src_major layouts (one chunk, read by phase B only) keep U and V at the SAME positions: no extra gaps
      const eoff_t al_c = (src_major || no_gaps) ? (eoff_t)pad : pick_align(cs[p.nchunks], p.nchunks);
      const eoff_t al_b = (src_major || no_gaps) ? (eoff_t)pad : pick_align(bs[p.nbins], p.nbins);
      std::vector<eoff_t> ca((size_t)p.nchunks + 1, 0), ba((size_t)p.nbins + 1, 0);
      for (unsigned c = 0; c < p.nchunks; c++) {
        du[c] = ca[c] - cs[c];
        ca[c + 1] = (ca[c] + (cs[c + 1] - cs[c]) + al_c - 1) & ~(al_c - 1);
      }
      for (unsigned b = 0; b < p.nbins; b++) {
        dv[b] = ba[b] - bs[b];
        ba[b + 1] = (ba[b] + (bs[b + 1] - bs[b]) + al_b - 1) & ~(al_b - 1);
      }
      n_pad = ca[p.nchunks] > ba[p.nbins] ? ca[p.nchunks] : ba[p.nbins];
      DevBuf<eoff_t> d_du, d_dv;
      GDN_TRY(d_du.alloc(p.nchunks));
      GDN_TRY(d_dv.alloc(p.nbins));
      GDN_HIP(hipMemcpy(d_du.p, du.data(), du.size() * sizeof(eoff_t), hipMemcpyHostToDevice));
      GDN_HIP(hipMemcpy(d_dv.p, dv.data(), dv.size() * sizeof(eoff_t), hipMemcpyHostToDevice));
      hipLaunchKernelGGL(pb_shift_kernel, dim3(gdn_nblocks(ntiles)), dim3(GDN_BLOCK), 0, 0, pu.p, pv.p, d_du.p, d_dv.p,
                         p.nchunks, p.nbins);
      GDN_HIP(hipMemcpy(p.chunk_ptr.p, ca.data(), ca.size() * sizeof(eoff_t), hipMemcpyHostToDevice));
      GDN_HIP(hipMemcpy(p.bin_ptr.p, ba.data(), ba.size() * sizeof(eoff_t), hipMemcpyHostToDevice));
      GDN_HIP(hipDeviceSynchronize());
    }
    p.n_pad = n_pad;
    if (gdn_xoption("GDN_PB_TRACE")) {
      eoff_t fill = 0;
      if (v_delta) (void)hipMemcpy(&fill, ds.p + n_use, sizeof(eoff_t), hipMemcpyDeviceToHost);
      fprintf(stderr, "[pb_build] edges %llu fillers %llu padded %llu (%.3f x) chunks %u bins %u pad %u group %u%s%s\n",
              (unsigned long long)n_use, (unsigned long long)fill, (unsigned long long)n_pad,
              n_use ? (double)n_pad / (double)n_use : 0.0, p.nchunks, p.nbins, pad, grp, v_delta ? " v8" : "",
              src_major ? " src-major" : "");
    }
    if ((n_pad >> log_group) + 1 > 0xFFFFFFFFull) {
      gdn_set_error("pb_build: more than 2^35 padded edges");
      return GDN_ERR_INVALID;
    }
    if (!early) {
      GDN_TRY(p.U.alloc(n_pad + grp));
      GDN_TRY(p.V.alloc(n_pad + grp));
      GDN_TRY(p.G.alloc((n_pad >> log_group) + 1));
    }
    const unsigned long long fb = (n_pad + grp + GDN_BLOCK - 1) / GDN_BLOCK;
    hipLaunchKernelGGL(pb_fill_u16_kernel, dim3((unsigned)(fb > 262144ull ? 262144ull : fb)), dim3(GDN_BLOCK), 0, 0, p.U.p,
                       n_pad + grp, (uint16_t)p.chunk_slots);  // pad edges read the zero slot behind the slice
    GDN_HIP(hipMemsetAsync(p.V.p, 0, (n_pad + grp) * sizeof(uint16_t), 0));
    float *ev_out = nullptr;
    if (edge_vals_in && edge_vals_out) {
      GDN_TRY(edge_vals_out->alloc(n_pad + grp));
      GDN_HIP(hipMemsetAsync(edge_vals_out->p, 0, (n_pad + grp) * sizeof(float), 0));
      ev_out = edge_vals_out->p;
    }
    if (n_use)
      hipLaunchKernelGGL(pb_scatter_kernel, dim3(grid_n), dim3(GDN_BLOCK), 0, 0, sorted, n_use, log_chunk, log_bin, bin_bits,
                         p.nchunks, p.nbins, tsu.p, pu.p, pv.p, p.U.p, p.V.p, g->rowptr, g->colidx,
                         ev_out ? edge_vals_in : nullptr, ev_out,
#ifdef GDN_EXPERIMENTS
                         gdn_xoption("GDN_PB_TEST_RANDV") ? 1 : 0,  // TIMING-ONLY experiment: uniform row ids
#else
                         0,
#endif
                         rows_are_sources ? 1 : 0, src_major ? 1 : 0, v_delta ? nd.p : nullptr, v_delta ? ds.p : nullptr,
                         (compact && ev_out) ? inv_s.p : nullptr, (compact && ev_out) ? inv_d.p : nullptr);
    if (v_delta) {  // pads repeat the tile's last row, then V (u16) -> one byte per edge + one u16 base per 32 edges
      hipLaunchKernelGGL(pb_pad_rows_kernel, dim3(gdn_nblocks(ntiles)), dim3(GDN_BLOCK), 0, 0, tsu.p, ds.p, pv.p, psz_b.p,
                         p.nchunks, p.nbins, p.V.p);
      GDN_TRY(p.Vd.alloc(n_pad + grp));
      GDN_TRY(p.Vb.alloc((n_pad >> 5) + 2));
      const unsigned long long eb = (n_pad + GDN_BLOCK - 1) / GDN_BLOCK;
      hipLaunchKernelGGL(pb_encode_rows_kernel, dim3((unsigned)(eb > 262144ull ? 262144ull : (eb ? eb : 1))), dim3(GDN_BLOCK),
                         0, 0, p.V.p, n_pad, p.Vd.p, p.Vb.p, p.errflag.p);
      GDN_HIP(hipGetLastError());
      unsigned bad = 0;
      GDN_HIP(hipMemcpy(&bad, p.errflag.p, sizeof(unsigned), hipMemcpyDeviceToHost));
      if (bad) {
        gdn_set_error("pb_build: a row distance of the delta-coded layout does not fit 8 bits (internal error)");
        return GDN_ERR_INVALID;
      }
      p.V.release();
      p.v8 = true;
    }
    int identity_g = 0;
#ifdef GDN_EXPERIMENTS  // GDN_PB_IDENTITY=1: TIMING-ONLY experiment (sequential phase-A stores, wrong results)
    identity_g = gdn_xoption("GDN_PB_IDENTITY") ? 1 : 0;
#endif
    {  // groups in the alignment gaps store their (neutral) values into the dump group behind the arrays
      const unsigned long long ng = (n_pad >> log_group) + 1;
      const unsigned long long fbg = (ng + GDN_BLOCK - 1) / GDN_BLOCK;
      hipLaunchKernelGGL(pb_fill_u32_kernel, dim3((unsigned)(fbg > 262144ull ? 262144ull : fbg)), dim3(GDN_BLOCK), 0, 0,
                         p.G.p, ng, (uint32_t)(n_pad >> log_group));
    }
    hipLaunchKernelGGL(pb_groups_kernel, dim3(gdn_nblocks(ntiles)), dim3(GDN_BLOCK), 0, 0, pu.p, pv.p, psz_c.p, p.nchunks,
                       p.nbins, p.G.p, identity_g, log_group);
    GDN_HIP(hipGetLastError());
    GDN_HIP(hipDeviceSynchronize());
  }  // key buffers freed here
  {  // largest-first launch order for both phases (hub chunks / bins would otherwise form the tail)
    std::vector<eoff_t> cp((size_t)p.nchunks + 1), bp((size_t)p.nbins + 1);
    GDN_HIP(hipMemcpy(cp.data(), p.chunk_ptr.p, cp.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
    GDN_HIP(hipMemcpy(bp.data(), p.bin_ptr.p, bp.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
#ifdef GDN_EXPERIMENTS  // GDN_PB_UNIFORM=1: TIMING-ONLY, equal-sized chunk / bin ranges (wrong results)
    if (gdn_xoption("GDN_PB_UNIFORM")) {
      const eoff_t al = atoi(gdn_xoption("GDN_PB_UNIFORM")) > 1 ? (eoff_t)atoi(gdn_xoption("GDN_PB_UNIFORM")) - 1 : 15;
      for (unsigned i = 0; i <= p.nchunks; i++) cp[i] = ((p.n_pad / p.nchunks) * i) & ~al;
      for (unsigned i = 0; i <= p.nbins; i++) bp[i] = ((p.n_pad / p.nbins) * i) & ~al;
      cp[p.nchunks] = bp[p.nbins] = p.n_pad;
      GDN_HIP(hipMemcpy(p.chunk_ptr.p, cp.data(), cp.size() * sizeof(eoff_t), hipMemcpyHostToDevice));
      GDN_HIP(hipMemcpy(p.bin_ptr.p, bp.data(), bp.size() * sizeof(eoff_t), hipMemcpyHostToDevice));
    }
#endif
    std::vector<uint32_t> co(p.nchunks), bo(p.nbins);
    for (unsigned i = 0; i < p.nchunks; i++) co[i] = i;
    for (unsigned i = 0; i < p.nbins; i++) bo[i] = i;
    std::stable_sort(co.begin(), co.end(), [&](uint32_t a, uint32_t b) { return cp[a + 1] - cp[a] > cp[b + 1] - cp[b]; });
    std::stable_sort(bo.begin(), bo.end(), [&](uint32_t a, uint32_t b) { return bp[a + 1] - bp[a] > bp[b + 1] - bp[b]; });
    GDN_TRY(p.chunk_order.alloc(p.nchunks));
    GDN_TRY(p.bin_order.alloc(p.nbins));
    GDN_HIP(hipMemcpy(p.chunk_order.p, co.data(), co.size() * 4, hipMemcpyHostToDevice));
    GDN_HIP(hipMemcpy(p.bin_order.p, bo.data(), bo.size() * 4, hipMemcpyHostToDevice));
  }
  if (alloc_vals) {
    if (!early) GDN_TRY(p.vals.alloc(p.n_pad + grp));
    GDN_HIP(hipMemset(p.vals.p, 0, (p.n_pad + grp) * sizeof(float)));
  }
  GDN_TRY(p.partial.alloc(p.nbins));
  GDN_TRY(p.red_scratch.alloc(2 * ((size_t)p.nbins / 4096 + 2)));
  GDN_HIP(hipDeviceSynchronize());
  if (gdn_xoption("GDN_PB_TRACE"))
    fprintf(stderr, "[pb_build] %.0f ms wall\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  return GDN_OK;
}

#include "gdn_pbtier.hpp"

// the builder above behind a linkable name (gdn_pr.hip, gdn_spmv.hip); see PbTieredArgs
int pb_build_tiered_run(const PbTieredArgs &a, PbPlan &p, PbTierSet &ts) { return pb_build_tiered(a, p, ts); }
int pb_build_out_tiered_run(const PbOutArgs &a, PbPlan &p, DevBuf<float> &Wp, PbOutTiers &ts) { return pb_build_out_tiered(a, p, Wp, ts); }

extern "C" {

int gdn_graph_transpose(const gdn_graph *g, gdn_graph **out) {
  GDN_REQUIRE(g != nullptr && out != nullptr, "graph / out");
  *out = nullptr;
  const int32_t m = g->m;
  DevBuf<unsigned long long> ka, kb, bigitems;
  DevBuf<unsigned> cnt;
  const uint64_t bigcap64 = g->nnz / EXP_CHUNK + (uint64_t)m / 64 + 1024;
  const unsigned bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
  GDN_TRY(ka.alloc_scratch(g->nnz));
  GDN_TRY(kb.alloc_scratch(g->nnz));
  GDN_TRY(bigitems.alloc(bigcap));
  GDN_TRY(cnt.alloc(2));
  GDN_HIP(hipMemset(cnt.p, 0, 8));
  ExpBigList big;
  big.items = bigitems.p;
  big.capacity = bigcap;
  big.count = cnt.p;
  big.overflow = cnt.p + 1;
  KeyVis vis;
  vis.colidx = g->colidx;
  vis.keys = ka.p;
  vis.v = 0;
  hipLaunchKernelGGL(transpose_keys_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, big,
                     vis);
  hipLaunchKernelGGL(transpose_keys_big_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
  GDN_HIP(hipGetLastError());
  unsigned h[2];
  GDN_HIP(hipMemcpy(h, cnt.p, 8, hipMemcpyDeviceToHost));
  if (h[1]) {
    gdn_set_error("gdn_graph_transpose: device worklist overflow");
    return GDN_ERR_OVERFLOW;
  }
  bigitems.release();
  return csr_from_keys(ka, kb, g->nnz, m, bits_for(m), out);
}

// (src,dst) pairs -> sort keys; symmetrize doubles them.  bad[0] is set when an id is outside [0,m)
__global__ void __launch_bounds__(GDN_BLOCK)
edge_keys_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ dst, unsigned long long n, int32_t m,
                 int symmetrize, unsigned long long *__restrict__ keys, unsigned *__restrict__ bad) {
  unsigned long long i = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * GDN_BLOCK;
  for (; i < n; i += stride) {
    const int32_t a = src[i], b = dst[i];
    if (a < 0 || a >= m || b < 0 || b >= m) {
      *bad = 1u;
      keys[i] = 0ull;  // (0,0) is a self loop: dropped
      if (symmetrize) keys[n + i] = 0ull;
      continue;
    }
    keys[i] = ((unsigned long long)(unsigned)a << 32) | (unsigned)b;
    if (symmetrize) keys[n + i] = ((unsigned long long)(unsigned)b << 32) | (unsigned)a;
  }
}

int gdn_graph_from_edges(int32_t m, uint64_t n_edges, const int32_t *src, const int32_t *dst, int32_t symmetrize,
                         gdn_graph **out) {
  GDN_REQUIRE(out != nullptr, "out");
  *out = nullptr;
  GDN_REQUIRE(m > 0, "m");
  GDN_REQUIRE(n_edges == 0 || (src != nullptr && dst != nullptr), "src / dst");
  GDN_TRY(gdn_require_device());
  const unsigned long long n = n_edges, nk = symmetrize ? 2 * n : n;
  DevBuf<unsigned long long> ka, kb;
  DevBuf<unsigned> bad;
  GDN_TRY(ka.alloc_scratch(nk));
  GDN_TRY(kb.alloc_scratch(nk));
  GDN_TRY(bad.alloc(1));
  GDN_HIP(hipMemset(bad.p, 0, sizeof(unsigned)));
  if (n) {
    DevBuf<int32_t> ds, dd;
    GDN_TRY(ds.alloc(n));
    GDN_TRY(dd.alloc(n));
    GDN_HIP(hipMemcpy(ds.p, src, n * sizeof(int32_t), hipMemcpyHostToDevice));
    GDN_HIP(hipMemcpy(dd.p, dst, n * sizeof(int32_t), hipMemcpyHostToDevice));
    const unsigned long long nb64 = (n + GDN_BLOCK - 1) / GDN_BLOCK;
    hipLaunchKernelGGL(edge_keys_kernel, dim3((unsigned)(nb64 > 262144ull ? 262144ull : nb64)), dim3(GDN_BLOCK), 0, 0, ds.p,
                       dd.p, n, m, symmetrize ? 1 : 0, ka.p, bad.p);
    GDN_HIP(hipGetLastError());
    unsigned h = 0;
    GDN_HIP(hipMemcpy(&h, bad.p, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (h) {
      gdn_set_error("gdn_graph_from_edges: a vertex id lies outside [0,%d)", m);
      return GDN_ERR_INVALID;
    }
  }
  return csr_from_keys(ka, kb, nk, m, bits_for(m), out);
}

// undirected closure of a resident graph: every edge in both directions, duplicates dropped (what the
// reference loader does with symmetrize = true, csr_graph.h:112-115 + fill_data)
int gdn_graph_symmetrize(const gdn_graph *g, gdn_graph **out) {
  GDN_REQUIRE(g != nullptr && out != nullptr, "graph / out");
  *out = nullptr;
  const int32_t m = g->m;
  DevBuf<unsigned long long> ka, kb, bigitems;
  DevBuf<unsigned> cnt;
  const uint64_t bigcap64 = g->nnz / EXP_CHUNK + (uint64_t)m / 64 + 1024;
  const unsigned bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
  GDN_TRY(ka.alloc_scratch(2 * g->nnz));
  GDN_TRY(kb.alloc_scratch(2 * g->nnz));
  GDN_TRY(bigitems.alloc(bigcap));
  GDN_TRY(cnt.alloc(2));
  GDN_HIP(hipMemset(cnt.p, 0, 8));
  ExpBigList big;
  big.items = bigitems.p;
  big.capacity = bigcap;
  big.count = cnt.p;
  big.overflow = cnt.p + 1;
  KeyVis vis;
  vis.colidx = g->colidx;
  vis.keys = ka.p;
  vis.fwd = ka.p + g->nnz;
  vis.v = 0;
  hipLaunchKernelGGL(transpose_keys_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, big,
                     vis);
  hipLaunchKernelGGL(transpose_keys_big_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
  GDN_HIP(hipGetLastError());
  unsigned h[2];
  GDN_HIP(hipMemcpy(h, cnt.p, 8, hipMemcpyDeviceToHost));
  if (h[1]) {
    gdn_set_error("gdn_graph_symmetrize: device worklist overflow");
    return GDN_ERR_OVERFLOW;
  }
  bigitems.release();
  return csr_from_keys(ka, kb, 2 * g->nnz, m, bits_for(m), out);
}

int gdn_rmat_build_ex(int32_t scale, uint64_t n_edges, double a, double b, double c, uint64_t seed, int32_t flags,
                      gdn_graph **out_csr, gdn_graph **in_csr) {
  GDN_REQUIRE(scale >= 1 && scale <= 30, "scale must be in [1,30]");
  GDN_REQUIRE(a > 0.0 && b >= 0.0 && c >= 0.0 && a + b + c <= 1.0, "quadrant probabilities");
  GDN_TRY(gdn_require_device());
  if (out_csr) *out_csr = nullptr;
  if (in_csr) *in_csr = nullptr;
  const unsigned long long n = n_edges;
  int32_t m = (int32_t)(1u << scale);
  const bool permute = (flags & GDN_RMAT_PERMUTE) != 0, compact = (flags & GDN_RMAT_COMPACT) != 0;
  // thresholds as in gardenia_amd/graphio.py: floor(p * 2^32) of the running sums (0.57 / 0.76 / 0.95 give RMAT_TA / _TAB / _TABC)
  auto thr = [](double p) { return p >= 1.0 ? 0xFFFFFFFFu : (unsigned)(p * 4294967296.0); };
  const unsigned t_a = thr(a), t_ab = thr(a + b), t_abc = thr(a + b + c);
  DevBuf<unsigned> bits;       // compact: ids with an edge
  DevBuf<eoff_t> word_base;    // compact: new id of the first set bit of every 32-id word
  int32_t m_new = m;
  for (int which = 0; which < 2; which++) {
    gdn_graph **dst = which == 0 ? out_csr : in_csr;
    if (!dst) continue;
    DevBuf<unsigned long long> ka, kb;
    GDN_TRY(ka.alloc_scratch(n ? n : 1));
    GDN_TRY(kb.alloc_scratch(n ? n : 1));
    unsigned nb = (unsigned)((n + GDN_BLOCK - 1) / GDN_BLOCK > 262144ull ? 262144ull : (n + GDN_BLOCK - 1) / GDN_BLOCK);
    if (nb == 0) nb = 1;
    hipLaunchKernelGGL(rmat_keys_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, (int)scale, n, (unsigned long long)seed,
                       (int)permute, which, ka.p, t_a, t_ab, t_abc);
    GDN_HIP(hipGetLastError());
    int rc = GDN_OK;
    if (compact) {
      if (!bits.p) {  // (both directions hold the same edges: the map is made once)
        const unsigned nwords = (unsigned)(((uint64_t)m + 31) / 32);
        DevBuf<unsigned> cnt;
        rc = bits.alloc_scratch((size_t)nwords + 1);
        if (rc == GDN_OK) rc = cnt.alloc_scratch((size_t)nwords + 1);
        if (rc == GDN_OK) rc = word_base.alloc_scratch((size_t)nwords + 2);
        if (rc == GDN_OK && hipMemsetAsync(bits.p, 0, ((size_t)nwords + 1) * 4, 0) != hipSuccess) rc = GDN_ERR_HIP;
        if (rc == GDN_OK) {
          hipLaunchKernelGGL(rmat_mark_ids_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, ka.p, n, bits.p);
          hipLaunchKernelGGL(rmat_word_counts_kernel, dim3(gdn_nblocks((uint64_t)nwords)), dim3(GDN_BLOCK), 0, 0, bits.p, nwords,
                             cnt.p);
          rc = gdn_exclusive_scan_u32_to_u64(cnt.p, word_base.p, (size_t)nwords, 0);
        }
        eoff_t live = 0;
        if (rc == GDN_OK && hipMemcpy(&live, word_base.p + nwords, sizeof(eoff_t), hipMemcpyDeviceToHost) != hipSuccess) rc = GDN_ERR_HIP;
        if (rc == GDN_OK) m_new = live > 0 ? (int32_t)live : 1;
      }
      if (rc == GDN_OK) {
        hipLaunchKernelGGL(rmat_relabel_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, ka.p, n, bits.p, word_base.p);
        GDN_HIP(hipGetLastError());
      }
    }
    if (rc == GDN_OK) rc = csr_from_keys(ka, kb, n, compact ? m_new : m, compact ? bits_for(m_new) : scale, dst);
    if (rc != GDN_OK) {
      if (out_csr && *out_csr) {
        gdn_graph_free(*out_csr);
        *out_csr = nullptr;
      }
      return rc;
    }
  }
  return GDN_OK;
}

int gdn_rmat_build_range(int32_t scale, uint64_t n_edges, double a, double b, double c, uint64_t seed, int32_t flags, int32_t v_lo,
                         int32_t v_hi, gdn_graph **in_rows, int32_t *d_out_degree_partial) {
  GDN_REQUIRE(scale >= 1 && scale <= 30, "scale must be in [1,30]");
  GDN_REQUIRE(a > 0.0 && b >= 0.0 && c >= 0.0 && a + b + c <= 1.0, "quadrant probabilities");
  GDN_REQUIRE((flags & ~GDN_RMAT_PERMUTE) == 0, "gdn_rmat_build_range: GDN_RMAT_PERMUTE is the only flag (compaction needs every range)");
  const int32_t m = (int32_t)(1u << scale);
  GDN_REQUIRE(in_rows != nullptr && v_lo >= 0 && v_lo < v_hi && v_hi <= m, "in_rows / 0 <= v_lo < v_hi <= 2^scale");
  GDN_TRY(gdn_require_device());
  *in_rows = nullptr;
  auto thr = [](double p) { return p >= 1.0 ? 0xFFFFFFFFu : (unsigned)(p * 4294967296.0); };
  const unsigned t_a = thr(a), t_ab = thr(a + b), t_abc = thr(a + b + c);
  const unsigned long long n = n_edges;
  unsigned nb = (unsigned)((n + GDN_BLOCK - 1) / GDN_BLOCK > 65536ull ? 65536ull : (n + GDN_BLOCK - 1) / GDN_BLOCK);
  if (nb == 0) nb = 1;
  DevBuf<unsigned long long> counter;
  GDN_TRY(counter.alloc_scratch(1));
  // pass 1: how many keys the range keeps; pass 2: the keys
  unsigned long long kept = 0;
  GDN_HIP(hipMemsetAsync(counter.p, 0, 8, 0));
  hipLaunchKernelGGL(rmat_range_keys_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, (int)scale, n, (unsigned long long)seed,
                     (int)((flags & GDN_RMAT_PERMUTE) != 0), t_a, t_ab, t_abc, (unsigned)v_lo, (unsigned)v_hi,
                     (unsigned long long *)nullptr, counter.p, 0ull);
  GDN_HIP(hipGetLastError());
  GDN_HIP(hipMemcpy(&kept, counter.p, 8, hipMemcpyDeviceToHost));
  DevBuf<unsigned long long> ka, kb;
  GDN_TRY(ka.alloc_scratch(kept ? kept : 1));
  GDN_TRY(kb.alloc_scratch(kept ? kept : 1));
  GDN_HIP(hipMemsetAsync(counter.p, 0, 8, 0));
  hipLaunchKernelGGL(rmat_range_keys_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, (int)scale, n, (unsigned long long)seed,
                     (int)((flags & GDN_RMAT_PERMUTE) != 0), t_a, t_ab, t_abc, (unsigned)v_lo, (unsigned)v_hi, ka.p, counter.p, kept);
  GDN_HIP(hipGetLastError());
  // rows keep their global ids through the sort (a key whose row equals its column is a self loop there), the range is cut
  // out of the offsets afterwards: 8 (m + 1) bytes of offsets beside the range's own edges
  gdn_graph *whole = nullptr;
  GDN_TRY(csr_from_keys(ka, kb, kept, m, scale, &whole));
  gdn_graph *rows = nullptr;
  const int rc = gdn_graph_slice_rows(whole, v_lo, v_hi, &rows);
  gdn_graph_free(whole);
  GDN_TRY(rc);
  if (d_out_degree_partial && rows->nnz) {
    hipLaunchKernelGGL(out_degree_add_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, rows->colidx, (unsigned long long)rows->nnz,
                       d_out_degree_partial);
    GDN_HIP(hipGetLastError());
    GDN_HIP(hipDeviceSynchronize());
  }
  *in_rows = rows;
  return GDN_OK;
}

int gdn_rmat_build(int32_t scale, int32_t edge_factor, uint64_t seed, int32_t permute, gdn_graph **out_csr,
                   gdn_graph **in_csr) {
  GDN_REQUIRE(scale >= 1 && scale <= 30, "scale must be in [1,30]");
  GDN_REQUIRE(edge_factor >= 1, "edge_factor");
  // Graph500's quadrants (include/generator.h:88-90): the thresholds are exactly RMAT_TA / RMAT_TAB / RMAT_TABC
  return gdn_rmat_build_ex(scale, (uint64_t)edge_factor << scale, 0.57, 0.19, 0.19, seed, permute ? GDN_RMAT_PERMUTE : 0, out_csr,
                           in_csr);
}

}  // extern "C"
