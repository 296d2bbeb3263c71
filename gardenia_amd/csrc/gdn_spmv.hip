// gdn_spmv.hip -- CSR SpMV, fp32, merge-based.
//
// Reference path: SpmvSolver (src/spmv/spmv.h:29); OpenMP src/spmv/omp_base.cc:7-42; CUDA
// src/spmv/base.cu:13 (thread per row), warp.cu:26 (warp per row), vector.cu:27 (sub-warp
// per row chosen by nnz/m, x through a texture), push.cu:11 (scatter with atomics).  One
// merge-path pass (gdn_mergepath.hpp) replaces all four: y[i] += sum_k Ax[k]*x[Aj[k]] over
// the rows of (Ap, Aj); products are rounded before the add like the reference's x86 build.
#include <string.h>

#include <stdlib.h>

#include "gdn_mergepath.hpp"
#include "gdn_pb.hpp"

struct gdn_spmv_plan {
  int layout = GDN_LAYOUT_CSR;
  int32_t m = 0;        // rows
  int32_t n_cols = 0;   // length of x (== m for a whole matrix, the global size for a row shard)
  uint64_t nnz = 0;
  MpPlan mp;            // GDN_LAYOUT_CSR
  PbPlan pb;            // GDN_LAYOUT_PB
  DevBuf<float> Axp;    // PB: Ax in chunk-major tile order (pads 0)
  // hub tier (gdn_pb.hpp): the edges of the highest-degree columns skip the per-edge value stream
  bool has_hub = false;
  PbPlan hub;
  unsigned n_hubs = 0;
  DevBuf<uint32_t> hub_ids;
  DevBuf<float> hub_val;  // x of the hub columns, refreshed per multiply
  DevBuf<float> hub_Ax;   // Ax of the hub edges in record order (pads 0)
  DevBuf<uint32_t> hub_rec;  // the hub layout as bin-major (hub index << 14 | row) records
  // mid tiers (gdn_pb.hpp): the next degree levels below the hubs, read by phase B as (record, Ax) pairs = 8 B/edge
  int n_mid_tiers = 0;
  struct MidTier {
    PbPlan layout;  // only bin_ptr is kept
    unsigned n = 0;
    DevBuf<uint32_t> ids, rec;
    DevBuf<float> val, Ax;
    bool il = false;  // rec / Ax: whole 256-record blocks lane-interleaved (phase B form 2)
  } mid[PB_MAX_MID];
  bool pattern = false;  // PB layout of a 0/1 matrix: no Ax stream
  DevBuf<unsigned> mx;  // PB: [0] bits of max|Ax|, [1] bits of max|x| (per call), [2] rows to recompute (per call), [3] max row length
  DevBuf<float> scale;  // PB: [0] = 2^shift, [1] = 2^-shift (per call)
  // PB: the rows phase B hands back because a product lost bits in the fixed-point conversion and the row's sum is too
  // small to hide it (gdn_pb.hpp PbTracksLossy) are recomputed from the CSR the plan was built on -- which therefore has
  // to stay alive as long as the plan (like d_Ax, which gdn_spmv_dev is handed per call)
  const gdn_graph *csr = nullptr;
  DevBuf<int32_t> repair_rows;  // m entries
  bool track_lossy = false;
};

struct SpmvOp {
  const float *__restrict__ Ax;
  const float *__restrict__ x;
  float *__restrict__ y;
  __device__ __forceinline__ float load(uint64_t j, vid_t col) const {
    return gdn_fmul(x[col], __builtin_nontemporal_load(Ax + j));
  }
  struct Pre {
    float y;
  };
  __device__ __forceinline__ Pre pre(int32_t row) const { return Pre{y[row]}; }
  __device__ __forceinline__ double fin(int32_t row, float sum, const Pre &p) const {
    // PB_REPAIR_ROW (PB plans with `track` only): y[row] stays as it is, the repair pass adds the row's sum.  Any other
    // NaN is a NaN sum (nan / inf in Ax or x on the CSR layout) and is stored like the reference's loop stores it.
    if (track && __float_as_uint(sum) == PB_REPAIR_ROW) repair(row);
    else y[row] = gdn_fadd(p.y, sum);
    return 0.0;
  }
  __device__ __forceinline__ double finish(int32_t row, float sum) const { return fin(row, sum, pre(row)); }
  bool vec_ok;  // y 16-byte aligned: pb_epilogue4
  struct Pre4 {
    pb_f32x4 y;
  };
  __device__ __forceinline__ Pre4 pre4(int32_t row) const { return Pre4{*reinterpret_cast<const pb_f32x4 *>(y + row)}; }
  __device__ __forceinline__ double fin4(int32_t row, const float (&sum)[4], const Pre4 &p) const {
    pb_f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (track && __float_as_uint(sum[c]) == PB_REPAIR_ROW) {
        repair(row + c);
        o[c] = p.y[c];
      } else {
        o[c] = gdn_fadd(p.y[c], sum[c]);
      }
    }
    *reinterpret_cast<pb_f32x4 *>(y + row) = o;
    return 0.0;
  }
  // PB layout: signed fixed point with a per-call power-of-two scale (gdn_pb.hpp).  A product below 2^23 units loses
  // bits in the conversion; where that could show in a row's result (|sum| < 2^32 units) phase B passes NaN and the row
  // goes onto the repair list instead of being written: spmv_repair_kernel adds its sum computed in fp32 like the
  // reference's loop.  Every other row's sum is the exact sum of its (at most 1 unit off) products, rounded once:
  // relative error <= (lossy products of the row) * 2^-32.
  static constexpr bool kTrackLossy = true;
  bool track;                   // false: pattern plans (delta PageRank sums signed deltas to an absolute tolerance)
  unsigned *repair_cnt;         // 1 counter
  int32_t *repair_rows;
  unsigned repair_cap;
  __device__ __forceinline__ unsigned long long to_fixed_lossy(float v, unsigned &bad, bool &lz) const {
    unsigned unfit = 0u;  // inf, nan or beyond the scale (the scale is made for the FINITE values): 0 is added and the
    const unsigned long long f = pb_to_fixed_signed(v, scale[0], unfit, &lz);  // row is recomputed (must_repair)
    lz = (lz || unfit) && track;
    if (!track) bad |= unfit;
    return f;
  }
  __device__ __forceinline__ bool must_repair(float v) const {
    return (__float_as_uint(v * scale[0]) & 0x7FFFFFFFu) >= 0x5E800000u;  // pb_to_fixed_signed's range test
  }
  __device__ __forceinline__ void repair(int32_t row) const {
    if (!repair_cnt) return;
    const unsigned pos = atomicAdd(repair_cnt, 1u);
    if (pos < repair_cap) repair_rows[pos] = row;
  }
  const float *__restrict__ scale;
  __device__ __forceinline__ unsigned long long to_fixed(float v, unsigned &bad) const {
    return pb_to_fixed_signed(v, scale[0], bad);
  }
  __device__ __forceinline__ float from_fixed(unsigned long long a, unsigned &) const {
    return gdn_fmul((float)(long long)a, scale[1]);
  }
};

// max |v| over an array, as float bits (non-negative floats order like unsigned integers)
__global__ void __launch_bounds__(GDN_BLOCK)
spmv_absmax_kernel(const float *__restrict__ v, size_t n, unsigned *__restrict__ out) {
  unsigned mx = 0;
  for (size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * GDN_BLOCK) {
    const unsigned b = __float_as_uint(v[i]) & 0x7FFFFFFFu;
    mx = (b > mx && b < 0x7F800000u) ? b : mx;  // finite values only (inf / nan products are repaired row by row)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t = __shfl_xor(mx, o, 64);
    mx = t > mx ? t : mx;
  }
  if (gdn_lane() == 0 && mx) atomicMax(out, mx);
}

__global__ void __launch_bounds__(GDN_BLOCK)
spmv_maxdeg_kernel(const eoff_t *__restrict__ rowptr, int32_t m, unsigned *__restrict__ out) {
  unsigned mx = 0;
  for (size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < (size_t)m; i += (size_t)gridDim.x * GDN_BLOCK) {
    const eoff_t d = rowptr[i + 1] - rowptr[i];
    const unsigned b = d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)d;
    mx = b > mx ? b : mx;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t = __shfl_xor(mx, o, 64);
    mx = t > mx ? t : mx;
  }
  if (gdn_lane() == 0 && mx) atomicMax(out, mx);
}

// scale = 2^shift with max|Ax| * max|x| * maxdeg * 2^shift < 2^61  (no host round trip)
__global__ void spmv_scale_kernel(const unsigned *__restrict__ mx, float *__restrict__ scale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float a = __uint_as_float(mx[0]), x = __uint_as_float(mx[1]);
  const float bound = a * x * (float)(mx[3] ? mx[3] : 1u);
  int e = 0;
  if (!(bound < 3.0e38f)) e = 128;  // the bound itself overflows fp32 (the maxima are finite): the coarsest scale;
                                    // ordinary products then lose bits and their rows are recomputed in fp32
  else if (bound > 0.0f) (void)frexpf(bound, &e);  // bound < 2^e
  int shift = 61 - e;
  if (shift > 120) shift = 120;
  if (shift < -120) shift = -120;
  scale[0] = ldexpf(1.0f, shift);
  scale[1] = ldexpf(1.0f, -shift);
}

// y[row] += SUM Ax[k] * x[Aj[k]] for the rows on the repair list: one wave per row, fp32 products and sums like the
// reference's loop (src/spmv/omp_base.cc:22-33), lane-strided partial sums folded by a wave reduction
__global__ void __launch_bounds__(GDN_BLOCK)
spmv_repair_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const float *__restrict__ Ax,
                   const float *__restrict__ x, float *__restrict__ y, const int32_t *__restrict__ rows,
                   const unsigned *__restrict__ count, unsigned cap) {
  unsigned n = *count;
  n = n > cap ? cap : n;
  const unsigned lane = gdn_lane();
  const unsigned nwaves = gridDim.x * GDN_WAVES_PER_BLOCK;
  for (unsigned i = blockIdx.x * GDN_WAVES_PER_BLOCK + (threadIdx.x >> 6); i < n; i += nwaves) {
    const int32_t row = rows[i];
    float acc = 0.0f;
    for (eoff_t k = rowptr[row] + lane; k < rowptr[row + 1]; k += 64) acc = gdn_fadd(acc, gdn_fmul(x[colidx[k]], Ax ? Ax[k] : 1.0f));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc = gdn_fadd(acc, __shfl_xor(acc, o, 64));
    if (lane == 0) y[row] = gdn_fadd(y[row], acc);
  }
}

extern "C" {

// slice sizes as for PageRank (pb_pick_log in gdn_pr.hip): whole matrices keep >= 2^9 slices before the compaction, row
// shards 2^10
static int spmv_pick_log(int64_t n, int max_log, int slices_log) {
  int lg = 10;
  while (lg < max_log && ((int64_t)1 << (lg + slices_log)) < n) lg++;
  return lg;
}

static int spmv_plan_place(gdn_spmv_plan *p, const float *d_Ax, int tries, double budget_ms);

static thread_local bool g_spmv_no_place = false;  // set by gdn_spmv around its own plan: one multiply does not pay for a search
int gdn_spmv_plan_create(const gdn_graph *csr, const float *d_Ax, int32_t layout, gdn_spmv_plan **plan) {
  GDN_REQUIRE(csr != nullptr, "csr");
  return gdn_spmv_plan_create_cols(csr, d_Ax, csr->m, layout, plan);
}

int gdn_spmv_plan_create_cols(const gdn_graph *csr, const float *d_Ax, int32_t n_cols, int32_t layout,
                              gdn_spmv_plan **plan) {
  GDN_REQUIRE(plan != nullptr, "plan");
  *plan = nullptr;
  GDN_REQUIRE(csr != nullptr, "csr");
  GDN_REQUIRE(n_cols >= 1, "n_cols");
  GDN_REQUIRE(layout == GDN_LAYOUT_CSR || layout == GDN_LAYOUT_PB || layout == GDN_LAYOUT_AUTO, "layout");
  if (layout == GDN_LAYOUT_AUTO) {
    const char *env = gdn_option("GDN_SPMV_LAYOUT");
    if (env && env[0] == 'c') layout = GDN_LAYOUT_CSR;
    else if (env && env[0] == 'p') layout = GDN_LAYOUT_PB;
    else layout = (d_Ax != nullptr && csr->nnz >= (1ull << 22)) ? GDN_LAYOUT_PB : GDN_LAYOUT_CSR;
  }
  // PB without values = the PATTERN matrix (every nonzero 1): no value stream in either phase (delta PageRank's pull)
  gdn_spmv_plan *p = new gdn_spmv_plan();
  p->pattern = layout == GDN_LAYOUT_PB && d_Ax == nullptr;
  p->layout = layout;
  p->m = csr->m;
  p->n_cols = n_cols;
  p->nnz = csr->nnz;
  p->csr = csr;
  int st;
  if (layout == GDN_LAYOUT_CSR) {
    st = mp_plan_build(p->mp, csr, 0);
  } else {
    // compacted like PageRank's layout (columns that occur / rows that have entries), rows of whole 128-byte lines;
    // GDN_PB_COMPACT=0 / GDN_PB_HUBS=0 switch the two refinements off (A/B measurements)
    int slices_log = csr->m == n_cols ? 9 : 10;
    if (const char *e = gdn_xoption("GDN_PB_SLICES_LOG")) slices_log = atoi(e) >= 6 && atoi(e) <= 12 ? atoi(e) : slices_log;  // tuning knob
    const int lc = spmv_pick_log(n_cols, PB_MAX_LOG_CHUNK, slices_log), lb = spmv_pick_log(csr->m, PB_MAX_LOG_BIN, slices_log);
    const char *ce = gdn_test_option("GDN_PB_COMPACT"), *he = gdn_test_option("GDN_PB_HUBS"), *ve = gdn_test_option("GDN_PB_V8");
    const bool compact = !(ce && ce[0] == '0');
    const bool v_delta = ve && ve[0] == '1';  // off by default, see gdn_pr.hip
    uint64_t hub_min_nnz = 1ull << 24;
    if (const char *e = gdn_test_option("GDN_PB_HUB_MIN_NNZ")) hub_min_nnz = strtoull(e, nullptr, 10);  // test knob
    DevBuf<uint8_t> cls;
    PbScratch scratch;  // the key buffers of the layout builds below
    st = GDN_OK;
    const char *me = gdn_test_option("GDN_PB_MID");  // number of mid tiers (0 switches them off; A/B measurements)
    int max_mid = me ? atoi(me) : 2;  // two here (PageRank takes PB_MAX_MID): an SpMV record carries its Ax, 8 B per edge
    if (max_mid < 0 || lb > PB_MID_ROW_BITS) max_mid = 0;
    DevBuf<uint32_t> mid_ids[PB_MAX_MID];
    unsigned n_mid[PB_MAX_MID] = {};
    // phase B's streams in lane-interleaved blocks (gdn_pb.hpp: PbPlan::v_il, PbMidArgs::form 2); GDN_PB_V_IL=0 / GDN_PB_REC_IL=0: plain
    const char *vie = gdn_test_option("GDN_PB_V_IL"), *rie = gdn_test_option("GDN_PB_REC_IL");
    const bool il_v = !(vie && vie[0] == '0') && !v_delta, il_streams = !(rie && rie[0] == '0');
    const bool want_tiers = compact && csr->nnz >= hub_min_nnz && !(he && he[0] == '0');
    // Round 4: the main layout, the record tiers and the values' places in both from ONE gather pass and LDS-staged splits
    // that carry Ax along (pb_build_tiered_run with edge_vals, gdn_pbtier.hpp) instead of one sort of 8-byte keys per layout
    // (pb_build + pb_mid_finish: RMAT-25 plan 0.23 s).  GDN_PB_BUILDER=old, delta-coded rows and uncompacted layouts keep the
    // old builder, and so does a shape outside the new one's limits (rc 1).
    bool built = false;
    {
      const char *be = gdn_option("GDN_PB_BUILDER");
      if (!(be && be[0] == 'o') && compact && !v_delta && lb <= PB_MID_ROW_BITS) {
        PbTieredArgs ta;
        PbTierSet ts;
        ta.rowptr = csr->rowptr;
        ta.colidx = csr->colidx;
        ta.m_raw = n_cols;
        ta.m_rows = csr->m;
        ta.m_global = n_cols;
        ta.nnz = csr->nnz;
        ta.src_count = nullptr;  // (column counts are not known here: exact marks + sampled degrees)
        ta.log_chunk = lc;
        ta.log_bin = lb;
        ta.pad = 32;
        ta.log_group = 5;
        ta.tiers = want_tiers;
        ta.max_mid = max_mid;
        ta.min16 = 0;  // the default floor of a mid tier (PB_MID_MIN_PER_BIN16), as pb_pick_tiers below
        ta.interleave = il_streams;
        ta.v_interleave = il_v;
        ta.edge_vals = p->pattern ? nullptr : d_Ax;
        ta.main_vals = &p->Axp;
        const int rc = pb_build_tiered_run(ta, p->pb, ts);
        if (rc < 0) st = rc;
        else if (rc == GDN_OK) {
          built = true;
          int k = 0;
          if (ts.n > 0 && ts.first_is_hub) {
            p->n_hubs = ts.t[0].n_src;
            p->hub_ids.take(ts.t[0].ids);
            p->hub_rec.take(ts.t[0].rec);
            if (!p->pattern) p->hub_Ax.take(ts.t[0].A);
            p->hub.bin_ptr.take(ts.t[0].bin_ptr);
            p->hub.nnz = ts.t[0].nnz;
            p->hub.nbins = p->pb.nbins;
            p->hub.nchunks = 1;
            st = p->hub_val.alloc(PB_HUB_SLOTS + 3);
            p->has_hub = true;
            k = 1;
          }
          for (int t = k; t < ts.n && st == GDN_OK; t++) {
            gdn_spmv_plan::MidTier &mt = p->mid[t - k];
            mt.n = ts.t[t].n_src;
            mt.ids.take(ts.t[t].ids);
            mt.rec.take(ts.t[t].rec);
            if (!p->pattern) mt.Ax.take(ts.t[t].A);
            mt.layout.bin_ptr.take(ts.t[t].bin_ptr);
            mt.layout.nnz = ts.t[t].nnz;
            mt.layout.nbins = p->pb.nbins;
            mt.il = ts.t[t].interleaved;
            st = mt.val.alloc((size_t)mt.n + 4);
            p->n_mid_tiers = t - k + 1;
          }
        }
      }
    }
    if (!built && st == GDN_OK && want_tiers)
      st = pb_pick_tiers(csr, n_cols, lb, cls, p->hub_ids, &p->n_hubs, max_mid, mid_ids, n_mid);
    if (!built && st == GDN_OK)
      st = pb_build(csr, n_cols, lc, lb, p->pb, true, d_Ax, &p->Axp, compact, false, /*pad=*/32, /*log_group=*/5,
                    (p->n_hubs || n_mid[0]) ? cls.p : nullptr, 0, false, v_delta, nullptr, 0, false, false, PB_MAX_LOG_BIN,
                    &scratch);
    if (!built && st == GDN_OK && il_v) st = pb_v_interleave(p->pb);
    if (!built && st == GDN_OK && p->n_hubs) {
      st = pb_build(csr, n_cols, PB_HUB_LOG, lb, p->hub, false, d_Ax, &p->hub_Ax, true, false, 16, 4, cls.p, 1, true, false,
                    nullptr, 0, false, false, PB_MAX_LOG_BIN, &scratch);
      if (st == GDN_OK && (p->hub.nchunks != 1 || p->hub.nbins != p->pb.nbins)) {
        gdn_set_error("gdn_spmv_plan_create: hub layout does not line up with the main layout");
        st = GDN_ERR_INVALID;
      }
      if (st == GDN_OK) st = p->hub_val.alloc(PB_HUB_SLOTS + 3);  // + the window behind the last slot
      if (st == GDN_OK) st = pb_mid_finish(p->hub, p->n_hubs, p->hub_rec, &p->hub_Ax);
      if (st == GDN_OK) p->has_hub = true;
    }
    for (int t = 0; !built && t < PB_MAX_MID && st == GDN_OK && n_mid[t]; t++) {
      gdn_spmv_plan::MidTier &mt = p->mid[t];
      st = pb_build(csr, n_cols, 15, lb, mt.layout, false, d_Ax, &mt.Ax, true, false, 16, 4, cls.p, 2 + t, true, false, nullptr, 0,
                    false, false, PB_MAX_LOG_BIN, &scratch);
      if (st == GDN_OK && mt.layout.nbins != p->pb.nbins) {
        gdn_set_error("gdn_spmv_plan_create: mid layout %d does not line up with the main layout", t);
        st = GDN_ERR_INVALID;
      }
      if (st == GDN_OK) st = pb_mid_finish(mt.layout, n_mid[t], mt.rec, p->pattern ? nullptr : &mt.Ax);
      if (st == GDN_OK && il_streams && mt.layout.nbins) {
        // the whole 256-record blocks of every bin's stream (records and factors) lane-interleaved: phase B form 2
        hipLaunchKernelGGL(pb_stream_interleave_kernel, dim3(mt.layout.nbins), dim3(GDN_BLOCK), 0, 0, mt.rec.p,
                           p->pattern ? nullptr : mt.Ax.p, mt.layout.bin_ptr.p);
        mt.il = true;
      }
      if (st == GDN_OK) st = mt.val.alloc((size_t)n_mid[t] + 4);
      if (st == GDN_OK) {
        mt.n = n_mid[t];
        mt.ids.take(mid_ids[t]);
        p->n_mid_tiers = t + 1;
      }
    }
    if (st == GDN_OK && (p->has_hub || p->n_mid_tiers)) {
      // launch order of phase B by all the bytes of a bin: main stream 6 B/edge, records 8 B, 8 B per row
      const eoff_t *tp[PB_MAX_REC_TIERS];
      int nt = 0;
      if (p->has_hub) tp[nt++] = p->hub.bin_ptr.p;
      for (int t = 0; t < p->n_mid_tiers; t++) tp[nt++] = p->mid[t].layout.bin_ptr.p;
      st = pb_order_bins_by_work(p->pb, nt, tp, 6.0, p->pattern ? 4.0 : 8.0, 8.0);
    }
    if (st == GDN_OK) st = p->mx.alloc(4);
    if (st == GDN_OK) st = p->scale.alloc(2);
    p->track_lossy = !p->pattern;
    if (st == GDN_OK && p->track_lossy) st = p->repair_rows.alloc((size_t)csr->m);
    if (st == GDN_OK) {
      (void)hipMemset(p->mx.p, 0, 16);
      if (p->pattern) {
        const float one = 1.0f;
        (void)hipMemcpy(p->mx.p, &one, 4, hipMemcpyHostToDevice);  // max |Ax| = 1
      } else if (csr->nnz) {
        hipLaunchKernelGGL(spmv_absmax_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, d_Ax, (size_t)csr->nnz, p->mx.p);
      }
      hipLaunchKernelGGL(spmv_maxdeg_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, csr->rowptr, csr->m, p->mx.p + 3);
      const int lds_a = (int)((sizeof(float) << p->pb.log_chunk) + 16);
      const int lds_b = (int)((sizeof(unsigned long long) << p->pb.log_bin) + ((size_t)1 << p->pb.log_bin) / 4);
      hipError_t e = hipFuncSetAttribute((const void *)pb_expand_scaled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a);
      if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)pb_accumulate_kernel<SpmvOp>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
      if (e != hipSuccess) {
        gdn_set_error("hipFuncSetAttribute(dynamic LDS): %s", hipGetErrorString(e));
        st = GDN_ERR_HIP;
      }
    }
  }
  if (st == GDN_OK && hipDeviceSynchronize() != hipSuccess) {
    gdn_set_error("gdn_spmv_plan_create: layout kernels failed: %s", hipGetErrorString(hipGetLastError()));
    st = GDN_ERR_HIP;
  }
  // placement search (PbPlacer, gdn_pb.hpp) from 3 x 2^28 non-zeros on: three multiplies on scratch vectors per candidate.
  // GDN_SPMV_PLACE=<tries per array> (0 = off)
  unsigned long long place_from = 3ull << 28;  // (as for PageRank, gdn_pr.hip: below, plans show no placement spread)
  if (const char *e = gdn_test_option("GDN_PLACE_MIN_EDGES")) place_from = strtoull(e, nullptr, 10);  // (tests force the search)
  if (st == GDN_OK && p->layout == GDN_LAYOUT_PB && csr->nnz >= place_from && !g_spmv_no_place) {
    int tries = 3;
    if (const char *e = gdn_option("GDN_SPMV_PLACE")) tries = atoi(e);
    if (tries > 0) st = spmv_plan_place(p, d_Ax, tries, 1000.0);
  }
  if (st != GDN_OK) {
    delete p;
    return st;
  }
  *plan = p;
  return GDN_OK;
}

static int spmv_plan_place(gdn_spmv_plan *p, const float *d_Ax, int tries, double budget_ms) {
  DevBuf<float> x, y;
  GDN_TRY(x.alloc((size_t)p->n_cols));
  GDN_TRY(y.alloc((size_t)p->m));
  const float half = 0.5f;
  GDN_TRY(gdn_fill_i32(reinterpret_cast<int32_t *>(x.p), __builtin_bit_cast(int32_t, half), (size_t)p->n_cols, 0));
  GDN_HIP(hipMemset(y.p, 0, (size_t)p->m * sizeof(float)));
  HostTimer t;
  PbPlacer pl;
  pl.tries = tries;
  pl.budget_ms = budget_ms;
  pl.tag = "spmv";
  pl.trace = gdn_option("GDN_SPMV_PLACE_TRACE") != nullptr;
  pl.timed = [&](double *out_ms) -> int {
    GDN_TRY(gdn_spmv_dev(p, d_Ax, x.p, y.p, nullptr));
    GDN_HIP(hipDeviceSynchronize());
    t.start();
    for (int k = 0; k < 3; k++) GDN_TRY(gdn_spmv_dev(p, d_Ax, x.p, y.p, nullptr));
    *out_ms = t.stop_ms() / 3.0;
    return GDN_OK;
  };
  GDN_TRY(pl.begin());
  int rc = pl.search(p->pb.vals, "vals", 2);
  if (rc == GDN_OK) rc = pl.search(p->pb.V, "V");
  if (rc == GDN_OK) rc = pl.search(p->Axp, "Ax");
  for (int k = 0; k < p->n_mid_tiers && rc == GDN_OK; k++) {
    rc = pl.search(p->mid[k].rec, "mid records");
    if (rc == GDN_OK) rc = pl.search(p->mid[k].Ax, "mid Ax");
  }
  if (rc == GDN_OK && p->has_hub) {
    rc = pl.search(p->hub_rec, "hub records");
    if (rc == GDN_OK) rc = pl.search(p->hub_Ax, "hub Ax");
  }
  if (rc == GDN_OK) rc = pl.search(p->pb.U, "U");
  pl.end();
  return rc;
}

int gdn_spmv_plan_free(gdn_spmv_plan *plan) {
  delete plan;
  return GDN_OK;
}

int gdn_spmv_dev(gdn_spmv_plan *plan, const float *d_Ax, const float *d_x, float *d_y, void *stream) {
  GDN_REQUIRE(plan && d_x && d_y, "null argument");
  GDN_REQUIRE(d_Ax != nullptr || plan->layout != GDN_LAYOUT_PB || plan->pattern,
              "d_Ax (the values in CSR order: rows that lose bits in the fixed-point layout are recomputed from them)");
  SpmvOp op;
  op.track = false;
  op.repair_cnt = nullptr;
  op.repair_rows = nullptr;
  op.repair_cap = 0;
  op.Ax = d_Ax;
  op.x = d_x;
  op.y = d_y;
  op.scale = nullptr;
  op.vec_ok = (reinterpret_cast<uintptr_t>(d_y) & 15u) == 0;
  hipStream_t s = (hipStream_t)stream;
  if (plan->layout == GDN_LAYOUT_CSR) {
    GDN_REQUIRE(d_Ax != nullptr, "d_Ax");
    return mp_run(plan->mp, op, nullptr, s);
  }
  // ---- propagation-blocked path (Ax lives in the plan in tile order; d_Ax is not read)
  PbPlan &pb = plan->pb;
  // max |x| (for the fixed-point scale of phase B) is collected by the launches that read x anyway: phase A over the
  // columns of the main layout, the gather kernels over the tier columns -- every column that has a nonzero
  GDN_HIP(hipMemsetAsync(plan->mx.p + 1, 0, 2 * sizeof(unsigned), s));  // max |x| and the repair count
  op.scale = plan->scale.p;
  op.track = plan->track_lossy;
  op.repair_cnt = plan->mx.p + 2;
  op.repair_rows = plan->repair_rows.p;
  op.repair_cap = (unsigned)plan->m;
  const size_t lds_a = (sizeof(float) << pb.log_chunk) + 16;
  const size_t lds_b = (sizeof(unsigned long long) << pb.log_bin) + ((size_t)1 << pb.log_bin) / 4;  // + the lossy / must-repair bitmaps
  const bool timed = pb.timing && pb.ev_used + 3 <= pb.ev.size();
  if (timed) GDN_HIP(hipEventRecord(pb.ev[pb.ev_used], s));
  hipLaunchKernelGGL(pb_expand_scaled_kernel, dim3(pb.nchunks), dim3(PB_THREADS), lds_a, s, d_x, pb.m_global,
                     pb.log_chunk, pb.chunk_ptr.p, pb.chunk_order.p, pb.U.p, pb.G.p, plan->pattern ? nullptr : plan->Axp.p, pb.vals.p,
                     pb.log_group, pb.compact ? pb.src_bits.p : nullptr, pb.compact ? pb.chunk_lo.p : nullptr,
                     pb.chunk_slots, plan->mx.p + 1);
  PbMidArgs mid = PbMidArgs();
  if (plan->has_hub) {
    hipLaunchKernelGGL(pb_tier_gather_f32_kernel, dim3(64), dim3(GDN_BLOCK), 0, s, d_x,
                       plan->hub_ids.p, plan->n_hubs, (unsigned)PB_HUB_SLOTS + 3u, plan->hub_val.p, plan->mx.p + 1);
    mid.ptr[mid.n] = plan->hub.bin_ptr.p;
    mid.rec[mid.n] = plan->hub_rec.p;
    mid.val[mid.n] = plan->hub_val.p;
    mid.A[mid.n] = plan->pattern ? nullptr : plan->hub_Ax.p;
    mid.zrec[mid.n] = plan->n_hubs << PB_MID_ROW_BITS;
    mid.form[mid.n++] = 1;
  }
  for (int t = 0; t < plan->n_mid_tiers; t++) {
    hipLaunchKernelGGL(pb_tier_gather_f32_kernel, dim3(128), dim3(GDN_BLOCK), 0, s,
                       d_x, plan->mid[t].ids.p, plan->mid[t].n, plan->mid[t].n + 4u, plan->mid[t].val.p, plan->mx.p + 1);
    mid.ptr[mid.n] = plan->mid[t].layout.bin_ptr.p;
    mid.rec[mid.n] = plan->mid[t].rec.p;
    mid.val[mid.n] = plan->mid[t].val.p;
    mid.A[mid.n] = plan->pattern ? nullptr : plan->mid[t].Ax.p;
    mid.zrec[mid.n] = plan->mid[t].n << PB_MID_ROW_BITS;
    mid.form[mid.n++] = plan->mid[t].il ? 2 : 0;
  }
  mid.v_il = pb.v_il ? 1 : 0;
  hipLaunchKernelGGL(spmv_scale_kernel, dim3(1), dim3(64), 0, s, plan->mx.p, plan->scale.p);
  if (timed) GDN_HIP(hipEventRecord(pb.ev[pb.ev_used + 1], s));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(pb_accumulate_kernel<SpmvOp>), dim3(pb.nbins), dim3(PB_THREADS), lds_b, s,
                     pb.m_local, pb.log_bin, pb.bin_ptr.p, pb.bin_order.p, pb.V.p, pb.vals.p, pb.partial.p,
                     pb.errflag.p, pb.compact ? pb.dst_bits.p : nullptr, pb.compact ? pb.bin_lo.p : nullptr, op, 0, 0u,
                     nullptr, nullptr, nullptr, nullptr, pb.v8 ? pb.Vd.p : nullptr, pb.v8 ? pb.Vb.p : nullptr, nullptr,
                     nullptr, nullptr, nullptr, mid);
  if (plan->track_lossy)  // the rows handed back (usually a handful or none: the kernel reads the count on the device)
    hipLaunchKernelGGL(spmv_repair_kernel, dim3(256), dim3(GDN_BLOCK), 0, s, plan->csr->rowptr, plan->csr->colidx, d_Ax, d_x, d_y,
                       plan->repair_rows.p, plan->mx.p + 2, (unsigned)plan->m);
  if (timed) {
    GDN_HIP(hipEventRecord(pb.ev[pb.ev_used + 2], s));
    pb.ev_used += 3;
  }
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

int gdn_spmv_plan_kernel_time(gdn_spmv_plan *plan, int32_t reset, int32_t max_launches, double *total_ms,
                              int32_t *launches) {
  GDN_REQUIRE(plan != nullptr, "plan");
  if (total_ms) total_ms[0] = total_ms[1] = 0.0;
  if (plan->layout == GDN_LAYOUT_CSR) return mp_plan_timing(plan->mp, reset, max_launches, total_ms, launches);
  PbPlan &pb = plan->pb;
  if (reset) {
    while (pb.ev.size() < (size_t)max_launches * 3) {
      hipEvent_t e;
      GDN_HIP(hipEventCreate(&e));
      pb.ev.push_back(e);
    }
    pb.ev_used = 0;
    pb.timing = max_launches > 0;
    return GDN_OK;
  }
  double a = 0, b = 0;
  for (size_t i = 0; i + 3 <= pb.ev_used; i += 3) {
    GDN_HIP(hipEventSynchronize(pb.ev[i + 2]));
    float ms = 0;
    GDN_HIP(hipEventElapsedTime(&ms, pb.ev[i], pb.ev[i + 1]));
    a += ms;
    GDN_HIP(hipEventElapsedTime(&ms, pb.ev[i + 1], pb.ev[i + 2]));
    b += ms;
  }
  if (total_ms) {
    total_ms[0] = a;
    total_ms[1] = b;
  }
  if (launches) *launches = (int32_t)(pb.ev_used / 3);
  pb.timing = false;
  return GDN_OK;
}

int gdn_spmv_plan_tiers(const gdn_spmv_plan *plan, int32_t *n_hubs, int32_t *n_mid_tiers, uint64_t *tier_edges) {
  GDN_REQUIRE(plan != nullptr, "plan");
  uint64_t ne = plan->has_hub ? plan->hub.nnz : 0;
  for (int t = 0; t < plan->n_mid_tiers; t++) ne += plan->mid[t].layout.nnz;
  if (n_hubs) *n_hubs = plan->has_hub ? (int32_t)plan->n_hubs : 0;
  if (n_mid_tiers) *n_mid_tiers = plan->n_mid_tiers;
  if (tier_edges) *tier_edges = ne;
  return GDN_OK;
}

// PB layout: GDN_ERR_OVERFLOW if a product left the fixed-point range (non-finite inputs)
int gdn_spmv_plan_check(gdn_spmv_plan *plan) {
  GDN_REQUIRE(plan != nullptr, "plan");
  if (plan->layout != GDN_LAYOUT_PB) return GDN_OK;
  unsigned f = 0;
  GDN_HIP(hipMemcpy(&f, plan->pb.errflag.p, sizeof(f), hipMemcpyDeviceToHost));
  if (f) {
    gdn_set_error("PB SpMV: a product overflowed the fixed-point accumulator (non-finite Ax or x?)");
    return GDN_ERR_OVERFLOW;
  }
  return GDN_OK;
}

// 8(m+1) + 4 nnz [Aj] + 4 nnz [Ax] + 4 nnz [x gather] + 8 m [y r+w]
uint64_t gdn_spmv_bytes(const gdn_spmv_plan *plan) {
  if (!plan) return 0;
  const uint64_t m = (uint64_t)plan->m, nnz = plan->nnz;
  return 8 * (m + 1) + (plan->pattern ? 8 : 12) * nnz + 8 * m;
}

// Host API: one call == SpmvSolver(g, Ax, x, y) (src/spmv/main.cc:39).
int gdn_spmv(int32_t m, uint64_t nnz, const uint64_t *Ap, const int32_t *Aj, const float *Ax,
             const float *x, float *y, gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && Ap && x && y && (Ax || nnz == 0), "null argument");
  GDN_TRY(gdn_require_device());
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  HostTimer th2d, tprep, tsolve;
  th2d.start();
  gdn_graph *g = nullptr;
  GDN_TRY(gdn_graph_upload(m, nnz, Ap, Aj, &g));
  DevBuf<float> d_Ax, d_x, d_y;
  gdn_spmv_plan *plan = nullptr;
  int rc = GDN_OK;
  do {
    if ((rc = d_Ax.alloc(nnz)) || (rc = d_x.alloc(m)) || (rc = d_y.alloc(m))) break;
    if ((nnz && hipMemcpy(d_Ax.p, Ax, nnz * 4, hipMemcpyHostToDevice) != hipSuccess) ||
        hipMemcpy(d_x.p, x, (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_y.p, y, (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess) {
      gdn_set_error("gdn_spmv: upload failed");
      rc = GDN_ERR_HIP;
      break;
    }
    st.h2d_ms = th2d.stop_ms();
    tprep.start();
    // One multiply: no layout build -- the merge-path pass over the caller's CSR, whose x gather runs at the L2-miss request
    // rate.  GDN_SPMV_ONESHOT=solve: the blocked layout instead, its build reported in prep_ms and outside
    // solve_ms -- the boundary of the reference's own blocked solver, whose segmenting() runs before its Timer starts
    // (src/spmv/partition.cu:206,269-291).  Wall time (prep + solve) is ~5x the default's; solve_ms is what the reference
    // would print.  The default stays the choice by wall time.
    int32_t layout = GDN_LAYOUT_CSR;
    if (const char *e = gdn_option("GDN_SPMV_ONESHOT"))
      if (e[0] == 's' && nnz >= (1ull << 22)) layout = GDN_LAYOUT_PB;
    const bool no_place = g_spmv_no_place;
    g_spmv_no_place = true;  // (one multiply does not pay for a placement search)
    rc = gdn_spmv_plan_create(g, layout == GDN_LAYOUT_PB ? d_Ax.p : nullptr, layout, &plan);
    g_spmv_no_place = no_place;
    if (rc) break;
    if (hipDeviceSynchronize() != hipSuccess) {
      gdn_set_error("gdn_spmv: layout build failed: %s", hipGetErrorString(hipGetLastError()));
      rc = GDN_ERR_HIP;
      break;
    }
    st.prep_ms = tprep.stop_ms();
    tsolve.start();  // src/spmv/warp.cu:100-104: one timed launch
    if ((rc = gdn_spmv_dev(plan, d_Ax.p, d_x.p, d_y.p, nullptr))) break;
    st.solve_ms = tsolve.stop_ms();
    st.iterations = 1;
    st.edges_traversed = nnz;
    if (hipMemcpy(y, d_y.p, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_spmv: download failed");
      rc = GDN_ERR_HIP;
    }
  } while (0);
  gdn_spmv_plan_free(plan);
  gdn_graph_free(g);
  if (stats) *stats = st;
  return rc;
}

}  // extern "C"
