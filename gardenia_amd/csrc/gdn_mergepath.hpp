// gdn_mergepath.hpp -- merge-based CSR row reduction for gfx950 (PageRank pull, SpMV).
//
// Supersedes the reference's thread-per-row / warp-per-row / sub-warp "vector" row kernels
// (src/pr/base.cu:19 pull_step, src/pr/warp.cu:32, src/spmv/base.cu:13 spmv_csr_scalar,
// src/spmv/warp.cu:26, src/spmv/vector.cu:27), whose work per thread follows the degree skew.
// Here the merged sequence {nonzeros, row-end markers} (m + nnz items) is cut into equal
// tiles of MP_TILE items, so every workgroup streams the same number of bytes no matter how
// skewed the degrees are:
//
//   tile_row[t]   = number of row ends among the first t*MP_TILE merged items (built once per
//                   graph by mp_tile_rows: one 27-step binary search per tile)
//   per tile      : coalesced, non-temporal stream of column_indices -> gather -> values
//                   staged in LDS (the "LDS-staged neighbour-list tile"); each thread then
//                   walks MP_IPT merged items serially; rows cut by thread boundaries are
//                   stitched by a wave64 segmented scan (shuffles) + 4-entry LDS hand-over.
//   rows cut by tile boundaries are finished by mp_fixup_kernel from per-tile carry/head
//   partials, in tile order: every sum has ONE fixed association, so results are bitwise
//   reproducible run to run (no float atomics anywhere).
//
// Row offsets are 64-bit end to end; everything tile-local is 16/32-bit.
#pragma once
#include <vector>

#include "gdn_common.hpp"

#define MP_IPT 16
#define MP_TILE (GDN_BLOCK * MP_IPT)  // 4096 merged items per workgroup

struct MpPlan {
  int32_t m = 0;
  uint64_t nnz = 0;
  const eoff_t *rowptr = nullptr;
  const vid_t *colidx = nullptr;
  uint64_t total_items = 0;
  uint32_t ntiles = 0;
  uint32_t nfix_blocks = 0;
  DevBuf<int32_t> tile_row;  // ntiles + 1
  DevBuf<float> tile_carry;  // ntiles: partial sum of the row still open at the tile's end
  DevBuf<float> tile_head;   // ntiles: partial sum of the tile's first row when it began earlier
  DevBuf<double> partial;    // ntiles + nfix_blocks (+ reduction scratch)
  DevBuf<double> red_scratch;
  // optional per-launch timing of the dominant kernel (mp_tile_kernel) with HIP events on the
  // launch stream; feeds bench.py's roofline.achieved
  bool timing = false;
  std::vector<hipEvent_t> ev;  // pairs
  size_t ev_used = 0;
  ~MpPlan() {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }
};

#ifdef __HIPCC__

static __global__ void __launch_bounds__(GDN_BLOCK)
mp_tile_rows(const eoff_t *__restrict__ rowptr, int32_t m, uint64_t total_items, uint32_t ntiles,
             int32_t *__restrict__ tile_row) {
  const uint32_t t = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (t > ntiles) return;
  uint64_t d = (uint64_t)t * MP_TILE;
  if (d > total_items) d = total_items;
  // row end k sits at merged position rowptr[k+1] + k; count those < d
  int32_t lo = 0, hi = m;
  while (lo < hi) {
    const int32_t mid = lo + ((hi - lo) >> 1);
    if (rowptr[mid + 1] + (uint64_t)mid < d) lo = mid + 1;
    else hi = mid;
  }
  tile_row[t] = lo;
}

// Op contract:
//   __device__ float  load(uint64_t j, vid_t col) const;       value of nonzero j
//   __device__ double finish(int32_t row, float sum) const;    consume a finished row sum,
//                                                              returns its L1 contribution
template <class Op>
__global__ void __launch_bounds__(GDN_BLOCK)
mp_tile_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx,
               const int32_t *__restrict__ tile_row, uint64_t total_items,
               float *__restrict__ tile_carry, float *__restrict__ tile_head,
               double *__restrict__ partial, Op op) {
  if (gdn_skip_launch(op)) return;
  __shared__ uint16_t s_rowend[MP_TILE + 2];
  __shared__ float s_val[MP_TILE + MP_TILE / 32];
  __shared__ float s_wave_v[GDN_WAVES_PER_BLOCK];
  __shared__ int s_wave_f[GDN_WAVES_PER_BLOCK];
  __shared__ double s_red[GDN_WAVES_PER_BLOCK];

  const uint32_t t = blockIdx.x;
  const int tid = threadIdx.x;
  const uint64_t d0 = (uint64_t)t * MP_TILE;
  const uint64_t d1 = (d0 + MP_TILE < total_items) ? d0 + MP_TILE : total_items;
  const int32_t i0 = tile_row[t], i1 = tile_row[t + 1];
  const uint64_t j0 = d0 - (uint64_t)i0;
  const int nrows = i1 - i0;
  const int nn = (int)((d1 - (uint64_t)i1) - j0);
  const int total = nrows + nn;

  // ---- stream the tile: row ends, then column indices -> gathered values in LDS
  for (int k = tid; k < nrows; k += GDN_BLOCK)
    s_rowend[k] = (uint16_t)(__builtin_nontemporal_load(rowptr + i0 + 1 + k) - j0);
  const vid_t *cbase = colidx + j0;
  vid_t cols[MP_IPT];
#pragma unroll
  for (int k = 0; k < MP_IPT; k++) {
    const int jj = k * GDN_BLOCK + tid;
    cols[k] = (jj < nn) ? __builtin_nontemporal_load(cbase + jj) : -1;
  }
  float vals[MP_IPT];
#pragma unroll
  for (int k = 0; k < MP_IPT; k++) {
    const int jj = k * GDN_BLOCK + tid;
    vals[k] = (cols[k] >= 0) ? op.load(j0 + (uint64_t)jj, cols[k]) : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < MP_IPT; k++) {
    const int jj = k * GDN_BLOCK + tid;
    if (jj < nn) s_val[jj + (jj >> 5)] = vals[k];
  }
  __syncthreads();

  // ---- per-thread serial merge over MP_IPT items
  const int dd = tid * MP_IPT;
  int ti, tj;
  if (dd < total) {
    int lo = dd - nn > 0 ? dd - nn : 0;
    int hi = dd < nrows ? dd : nrows;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((int)s_rowend[mid] + mid < dd) lo = mid + 1;
      else hi = mid;
    }
    ti = lo;
    tj = dd - lo;
  } else {
    ti = nrows;
    tj = nn;
  }
  const int ti_start = ti;
  float outv[MP_IPT];
  unsigned emit_mask = 0u;
  float running = 0.0f;
  unsigned re = (ti < nrows) ? (unsigned)s_rowend[ti] : 0xFFFFu;
#pragma unroll
  for (int s = 0; s < MP_IPT; s++) {
    outv[s] = 0.0f;
    if (dd + s < total) {
      if ((unsigned)tj < re) {
        running = gdn_fadd(running, s_val[tj + (tj >> 5)]);
        tj++;
      } else {
        outv[s] = running;
        emit_mask |= 1u << s;
        running = 0.0f;
        ti++;
        re = (ti < nrows) ? (unsigned)s_rowend[ti] : 0xFFFFu;
      }
    }
  }

  // ---- stitch rows cut by thread boundaries: segmented inclusive scan of the open partials
  //      (a thread that emitted a row starts a new segment)
  const unsigned lane = gdn_lane();
  const unsigned w = (unsigned)tid >> 6;
  float v = running;
  int f = emit_mask != 0u;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float pv = __shfl_up(v, o, 64);
    const int pf = __shfl_up(f, o, 64);
    if (lane >= (unsigned)o) {
      if (!f) v = gdn_fadd(pv, v);
      f |= pf;
    }
  }
  if (lane == 63) {
    s_wave_v[w] = v;
    s_wave_f[w] = f;
  }
  __syncthreads();  // also: every thread is done reading s_val
  float open_prev = 0.0f;
  for (unsigned ww = 0; ww < w; ww++) open_prev = s_wave_f[ww] ? s_wave_v[ww] : gdn_fadd(open_prev, s_wave_v[ww]);
  const float open = f ? v : gdn_fadd(open_prev, v);
  float carry_in = __shfl_up(open, 1, 64);
  if (lane == 0) carry_in = open_prev;
  if (tid == GDN_BLOCK - 1) tile_carry[t] = open;

  // ---- finished row sums -> LDS (aliases s_val), first one takes the carry-in
  {
    int r = ti_start;
    bool first = true;
#pragma unroll
    for (int s = 0; s < MP_IPT; s++) {
      if (emit_mask & (1u << s)) {
        s_val[r] = first ? gdn_fadd(carry_in, outv[s]) : outv[s];
        first = false;
        r++;
      }
    }
  }
  __syncthreads();

  // ---- epilogue over the rows that end in this tile (coalesced by row id)
  const bool has_head = (nrows > 0) && (rowptr[i0] < j0);
  double dsum = 0.0;
  for (int k = tid; k < nrows; k += GDN_BLOCK) {
    const float sum = s_val[k];
    if (k == 0 && has_head) tile_head[t] = sum;
    else dsum += op.finish(i0 + k, sum);
  }
  dsum = gdn_block_sum(dsum, s_red);
  if (tid == 0) partial[t] = dsum;
}

// Rows whose nonzeros straddle tile boundaries: one thread per tile that begins inside a row.
template <class Op>
__global__ void __launch_bounds__(GDN_BLOCK)
mp_fixup_kernel(const eoff_t *__restrict__ rowptr, const int32_t *__restrict__ tile_row,
                uint32_t ntiles, const float *__restrict__ tile_carry,
                const float *__restrict__ tile_head, double *__restrict__ partial_out, Op op) {
  if (gdn_skip_launch(op)) return;
  __shared__ double s_red[GDN_WAVES_PER_BLOCK];
  const uint32_t t = blockIdx.x * GDN_BLOCK + threadIdx.x;
  double d = 0.0;
  if (t < ntiles && t > 0) {
    const int32_t i0 = tile_row[t], i1 = tile_row[t + 1];
    if (i1 > i0) {
      const uint64_t j0 = (uint64_t)t * MP_TILE - (uint64_t)i0;
      if (rowptr[i0] < j0) {
        uint32_t ts = t - 1;
        while (ts > 0 && tile_row[ts] == i0) ts--;
        float sum = 0.0f;
        for (uint32_t tt = ts; tt < t; tt++) sum = gdn_fadd(sum, tile_carry[tt]);
        sum = gdn_fadd(sum, tile_head[t]);
        d = op.finish(i0, sum);
      }
    }
  }
  d = gdn_block_sum(d, s_red);
  if (threadIdx.x == 0) partial_out[blockIdx.x] = d;
}

// Deterministic tree reduction of doubles: chunk c of MP_RED_CHUNK inputs -> out[c].
#define MP_RED_CHUNK (GDN_BLOCK * 16)
static __global__ void __launch_bounds__(GDN_BLOCK)
mp_reduce_f64(const double *__restrict__ in, uint32_t n, double *__restrict__ out) {
  __shared__ double s_red[GDN_WAVES_PER_BLOCK];
  const uint32_t base = blockIdx.x * MP_RED_CHUNK;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const uint32_t i = base + k * GDN_BLOCK + threadIdx.x;
    if (i < n) acc += in[i];
  }
  acc = gdn_block_sum(acc, s_red);
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

static inline int mp_plan_build(MpPlan &p, const gdn_graph *g, hipStream_t s) {
  p.m = g->m;
  p.nnz = g->nnz;
  p.rowptr = g->rowptr;
  p.colidx = g->colidx;
  p.total_items = (uint64_t)g->m + g->nnz;
  const uint64_t nt = (p.total_items + MP_TILE - 1) / MP_TILE;
  if (nt > 0x7FFFFFFFull) {
    gdn_set_error("merge-path: too many tiles");
    return GDN_ERR_INVALID;
  }
  p.ntiles = (uint32_t)(nt == 0 ? 1 : nt);
  p.nfix_blocks = gdn_nblocks(p.ntiles);
  GDN_TRY(p.tile_row.alloc((size_t)p.ntiles + 1));
  GDN_TRY(p.tile_carry.alloc(p.ntiles));
  GDN_TRY(p.tile_head.alloc(p.ntiles));
  GDN_TRY(p.partial.alloc((size_t)p.ntiles + p.nfix_blocks));
  GDN_TRY(p.red_scratch.alloc(2 * ((size_t)(p.ntiles + p.nfix_blocks) / MP_RED_CHUNK + 2)));
  hipLaunchKernelGGL(mp_tile_rows, dim3(gdn_nblocks((uint64_t)p.ntiles + 1)), dim3(GDN_BLOCK), 0, s,
                     p.rowptr, p.m, p.total_items, p.ntiles, p.tile_row.p);
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

// reset != 0: (re)arm per-launch timing with room for max_launches; reset == 0: wait for the
// recorded events and report the summed duration of the dominant kernel and the launch count.
static inline int mp_plan_timing(MpPlan &p, int reset, int max_launches, double *total_ms, int32_t *launches) {
  if (reset) {
    while (p.ev.size() < (size_t)max_launches * 2) {
      hipEvent_t e;
      GDN_HIP(hipEventCreate(&e));
      p.ev.push_back(e);
    }
    p.ev_used = 0;
    p.timing = max_launches > 0;
    return GDN_OK;
  }
  double tot = 0;
  for (size_t i = 0; i + 1 < p.ev_used; i += 2) {
    GDN_HIP(hipEventSynchronize(p.ev[i + 1]));
    float ms = 0;
    GDN_HIP(hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]));
    tot += ms;
  }
  if (total_ms) total_ms[0] = tot;
  if (launches) *launches = (int32_t)(p.ev_used / 2);
  p.timing = false;
  return GDN_OK;
}

// One pass over the graph with Op; d_out (nullable) receives the reduced double.
template <class Op>
static inline int mp_run(MpPlan &p, const Op &op, double *d_out, hipStream_t s) {
  const bool timed = p.timing && p.ev_used + 2 <= p.ev.size();
  if (timed) GDN_HIP(hipEventRecord(p.ev[p.ev_used], s));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(mp_tile_kernel<Op>), dim3(p.ntiles), dim3(GDN_BLOCK), 0, s, p.rowptr,
                     p.colidx, p.tile_row.p, p.total_items, p.tile_carry.p, p.tile_head.p, p.partial.p, op);
  if (timed) {
    GDN_HIP(hipEventRecord(p.ev[p.ev_used + 1], s));
    p.ev_used += 2;
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(mp_fixup_kernel<Op>), dim3(p.nfix_blocks), dim3(GDN_BLOCK), 0, s,
                     p.rowptr, p.tile_row.p, p.ntiles, p.tile_carry.p, p.tile_head.p,
                     p.partial.p + p.ntiles, op);
  if (d_out) {
    uint32_t n = p.ntiles + p.nfix_blocks;
    const double *in = p.partial.p;
    double *bufs[2] = {p.red_scratch.p, p.red_scratch.p + (p.red_scratch.n / 2)};
    int which = 0;
    for (;;) {
      const uint32_t nb = (n + MP_RED_CHUNK - 1) / MP_RED_CHUNK;
      double *out = (nb == 1) ? d_out : bufs[which];
      hipLaunchKernelGGL(mp_reduce_f64, dim3(nb), dim3(GDN_BLOCK), 0, s, in, n, out);
      if (nb == 1) break;
      in = out;
      n = nb;
      which ^= 1;
    }
  }
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}
#endif  // __HIPCC__
