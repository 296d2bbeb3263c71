// gdn_pr.hip -- PageRank, pull direction, one fused pass per iteration.
//
// Reference path: PRSolver (src/pr/pr.h:31); OpenMP src/pr/omp_base.cc:8-42; CUDA
// src/pr/base.cu:77-135 = three launches per iteration (contrib :14, pull_step :19,
// l1norm :37) + a blocking 4-byte D2H of `diff` (:124).  Here one merge-path pass
// (gdn_mergepath.hpp) does gather + score update + L1 norm + next iteration's contrib;
// contrib is double-buffered so the iteration stays Jacobi like omp_base.cc:23-33.
// Arithmetic follows the reference in fp32: base_score = (1-d)/m, contrib = score/out_degree,
// new = base + d*sum (no FMA contraction), error accumulated in double (omp_base.cc:22).
#include <string.h>

#include <algorithm>
#include <vector>

#include <stdlib.h>

#include "gdn_mergepath.hpp"
#include <math.h>

#include "gdn_pb.hpp"
#include "gdn_seqsum.hpp"

int gdn_radix_sort_u64(unsigned long long *a, unsigned long long *b, unsigned long long n, unsigned begin_bit, unsigned end_bit,
                       const unsigned long long **sorted);  // gdn_sort.hip

struct gdn_pr_plan {
  bool placing = false;  // the placement search is running: its sweeps are launched under tagged kernel names
  const unsigned *skip_flag = nullptr;  // device word (gdn_pr's batched loop): non-zero = the pull launches do nothing
  int layout = GDN_LAYOUT_CSR;
  MpPlan mp;  // GDN_LAYOUT_CSR
  PbPlan pb;  // GDN_LAYOUT_PB
  // hub tier of the PB layout: the edges of the n_hubs sources with the most out-edges live in a second layout
  // (one source chunk, tiles sorted by hub) that phase B reads directly -- they never pass through vals
  bool has_hub = false;
  PbPlan hub;
  unsigned n_hubs = 0;
  DevBuf<uint32_t> hub_ids;  // original id of hub k (ascending)
  DevBuf<float> hub_val;     // PB_HUB_SLOTS values per iteration: contrib of hub k, slots >= n_hubs = 0 for pad edges
  DevBuf<uint32_t> hub_rec;  // the hub layout as bin-major (hub index << 14 | row) records (pb_mid_finish)
  // mid tiers (gdn_pb.hpp): the next degree levels below the hubs, read by phase B as 32-bit (source, row) records
  int n_mid_tiers = 0;
  struct MidTier {
    PbPlan layout;           // only bin_ptr is kept (pb_mid_finish)
    unsigned n = 0;          // sources
    DevBuf<uint32_t> ids;    // original id of source k (ascending)
    DevBuf<uint32_t> rec;    // bin-major records
    DevBuf<float> val;       // n + 1 values per iteration
    bool il = false;         // rec in lane-interleaved blocks of 256 (PbTierSet::Tier::interleaved): phase B form 2
  } mid[PB_MAX_MID];
  // hub-ROW tier: the edges into the n_hr rows with the most in-edges (from non-hub sources) are summed by phase A in
  // LDS behind the chunk's slice; one partial sum per (chunk, hub row) replaces one value per edge
  bool has_hr = false;
  PbPlan hr;                              // chunks as in `pb`, ONE bin over the hub rows; U and V share one order
  unsigned n_hr = 0;
  DevBuf<uint32_t> hr_ids;                // original row of hub row k (ascending)
  DevBuf<unsigned long long> hr_partial;  // nchunks x n_hr
  DevBuf<unsigned long long> hr_total;    // n_hr, per iteration
  DevBuf<unsigned> hrb_ptr;               // nbins + 1: hub rows of every bin (they are sorted by id = by bin)
  DevBuf<uint16_t> hrb_vl;                // n_hr: row index inside its bin
  int32_t m_local = 0;
  uint64_t nnz = 0;
  const int32_t *out_degree = nullptr;  // device, m_local
  int32_t m_global = 0;
  int32_t row_base = 0;
  int32_t m_base = 0;  // vertex count of the base score (1 - d) / m: the ORIGINAL m of a squished plan
  // GDN_LAYOUT_PB_SQUISHED: the plan works in a vertex space WITHOUT the vertices that have neither in- nor out-edges
  // (state index k <-> original id sq_ids[k], ascending).  Such a vertex keeps the constant base score and nobody reads
  // its contribution, so the per-vertex arrays of an iteration (scores, contrib, degrees) shrink to the m_state live
  // vertices; gdn_pr_import_dev / gdn_pr_export_dev convert at the boundary of a solve.
  bool squished = false;
  int32_t m_orig = 0;
  DevBuf<uint32_t> sq_ids;    // m_state
  DevBuf<uint32_t> sq_bits;   // ceil(m_orig / 32): live vertices
  DevBuf<int32_t> sq_deg;     // out-degrees in state order
  DevBuf<eoff_t> sq_rowptr;   // the relabelled in-CSR (kept while the plan lives only for the CSR of state rows)
  DevBuf<vid_t> sq_colidx;
  DevBuf<double> sq_diff;     // 1 double: L1 change of the dead vertices at the last import
  // GDN_PR_SUM=reference (diagnostic): behind every pull the rows with >= ref_min_deg in-edges are summed
  // AGAIN the way src/pr/omp_base.cc:27-33 sums them -- one fp32 add per in-edge, in CSR order, from the caller's in-CSR --
  // and scores / next contrib / L1 change are rewritten from those sums.  With ref_min_deg = 0 every row is: the iteration
  // then has the reference's bits, whatever layout the plan streams (pr_refsum_kernel).
  bool ref_sum = false;
  uint32_t ref_min_deg = 0;
  // (the selected rows, longest first; their entries chunk-major as (local source id, place in hv); hv = the entries' contributions
  // in row order, rewritten by the stage pass of every pull; per-row sums)
  DevBuf<uint32_t> ref_row, ref_deg, ref_cdest, ref_sumbits;
  DevBuf<uint16_t> ref_cu16;
  DevBuf<eoff_t> ref_off, ref_cptr;
  DevBuf<float> ref_hv;
  uint32_t ref_n = 0, ref_n_vlong = 0, ref_longest = 0, ref_nchunks = 0;
  uint64_t ref_edges = 0;
  hipStream_t ref_stream = nullptr;    // the very long rows' workgroups run beside the other rows' waves (pr_ref_resum)
  hipEvent_t ref_ev[2] = {nullptr, nullptr};
  ~gdn_pr_plan() {
    for (hipEvent_t e : ref_ev)
      if (e) (void)hipEventDestroy(e);
    if (ref_stream) (void)hipStreamDestroy(ref_stream);
  }
  DevBuf<float> ref_old;               // the selected rows' scores in front of the pull (the L1 change is corrected against them)
  DevBuf<double> ref_partial;          // per-workgroup corrections of the L1 change
  // gdn_pr_pull_parts_dev: one iteration = one launch per phase, its bins in part-major order with tickets (gdn_pb.hpp PbParts)
  DevBuf<unsigned> tickets;            // PB_MAX_PARTS counters, PB_TICKET_STRIDE words apart, + 1 timeout word behind them
  uint32_t ticket_target[PB_MAX_PARTS] = {};
  int32_t ticket_parts = 0;            // parts of the last ticketed pull
  std::vector<int32_t> parts_key;      // the row ends parts_order was made for
  DevBuf<uint32_t> parts_order;        // nbins: the bins of part 0 first, largest first inside a part
  PbParts parts_launch;                // launch-index end of every part
};

struct PrOp {
  static constexpr bool kSkippable = true;
  const unsigned *skip = nullptr;  // GdnSkippable: set while gdn_pr queues iterations in batches
  const float *__restrict__ contrib_in;
  float *__restrict__ scores;
  float *__restrict__ contrib_out;  // already offset by row_base
  const int32_t *__restrict__ out_degree;
  float base_score;
  float damping;
  __device__ __forceinline__ float load(uint64_t, vid_t col) const { return contrib_in[col]; }
  // PB layout: unsigned 2^-62 fixed point (gdn_pb.hpp).  What phase B reads (vals, the tier tables) are the CODES
  // pb_encode made of the contributions, once per source: per edge only the decode is left
  __device__ __forceinline__ unsigned long long to_fixed(float v, unsigned &) const { return pb_decode(__float_as_uint(v)); }
  __device__ __forceinline__ float from_fixed(unsigned long long a, unsigned &bad) const {
    if (a >> 63) bad = 1u;
    return ldexpf((float)a, -PB_FIX_SHIFT);
  }
  struct Pre {
    float old_score;
    int32_t deg;
  };
  __device__ __forceinline__ Pre pre(int32_t row) const { return Pre{scores[row], out_degree[row]}; }
  __device__ __forceinline__ double fin(int32_t row, float sum, const Pre &p) const {
    const float new_score = gdn_fadd(base_score, gdn_fmul(damping, sum));
    scores[row] = new_score;
    if (wt) __hip_atomic_store(contrib_out + row, __fdiv_rn(new_score, (float)p.deg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else contrib_out[row] = __fdiv_rn(new_score, (float)p.deg);
    return (double)fabsf(gdn_fsub(new_score, p.old_score));
  }
  __device__ __forceinline__ double finish(int32_t row, float sum) const { return fin(row, sum, pre(row)); }
  // 16-byte row accesses of the PB epilogue (pb_epilogue4); vec_ok = all three row arrays 16-byte aligned
  bool vec_ok;
  struct Pre4 {
    pb_f32x4 old_score;
    pb_i32x4 deg;
  };
  __device__ __forceinline__ Pre4 pre4(int32_t row) const {
    return Pre4{*reinterpret_cast<const pb_f32x4 *>(scores + row), *reinterpret_cast<const pb_i32x4 *>(out_degree + row)};
  }
  __device__ __forceinline__ double fin4(int32_t row, const float (&sum)[4], const Pre4 &p) const {
    pb_f32x4 ns, nc;
    double d = 0.0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const float new_score = gdn_fadd(base_score, gdn_fmul(damping, sum[c]));
      ns[c] = new_score;
      nc[c] = __fdiv_rn(new_score, (float)p.deg[c]);
      d += (double)fabsf(gdn_fsub(new_score, p.old_score[c]));
    }
    *reinterpret_cast<pb_f32x4 *>(scores + row) = ns;
    if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(contrib_out + row), "v"(nc) : "memory");
    else *reinterpret_cast<pb_f32x4 *>(contrib_out + row) = nc;
    return d;
  }
  // ticketed pulls (gdn_pr_pull_parts_dev): the next contributions -- what the ranks exchange -- are stored WRITE-THROUGH
  // (sc1: the line leaves the XCD's L2 at once), so that a workgroup's ticket needs no L2 write-back in front of it
  int wt = 0;
};

// ---- GDN_PR_SUM=reference: the reference's summation order on demand: the rows of >= ref_min_deg in-edges
// ("selected" rows) are summed AGAIN the way src/pr/omp_base.cc:27-30 sums them -- one fp32 addition per in-edge, in CSR
// order -- and their scores / next contributions / the L1 change rewritten from those sums.  Round 6, two passes per pull:
//   STAGE  (pr_ref_stage_kernel) the contributions the selected rows read, fetched the way phase A fetches: a workgroup per
//          chunk of 2^15 sources loads the chunk's slice of contrib_in into LDS (coalesced) and writes slice[local id] to the
//          place of every selected-row entry whose source lies in the chunk -- hv, in ROW order (the plan keeps the entries
//          chunk-major as (u16 local id, u32 place), built once).  No gather leaves the CU: a 4-byte gather from HBM / the
//          Infinity Cache pulls a 64-byte line, 24.8 GB per iteration for RMAT-27's rows of >= 10^4 in-edges.
//   SCAN   (pr_refscan_kernel / pr_refscan_wg_kernel) every row streams its values from hv and sums them in order -- not by
//          a chain of dependent additions but by scans of parity functions (gdn_seqsum.hpp: the same bits): a wave per row,
//          a WORKGROUP per very long row (16 waves reduce 16 blocks to pairs at once, one pass chains them: a lone wave pays
//          a block's latency once per block, and RMAT-27's longest row has 1 763 of them).
#define PR_REF_N 8                       // elements per lane and block
#define PR_REF_BLOCK (64 * PR_REF_N)     // 512 elements per wave step
#define PR_REF_LOG_CHUNK 15              // sources per staging chunk (128 KB of LDS)
#define PR_REF_STAGE_THREADS 1024
struct PrRefRows {
  const uint32_t *__restrict__ row;   // state row of selected row i (sorted by in-degree, descending)
  const uint32_t *__restrict__ deg;   // its in-degree
  const eoff_t *__restrict__ off;     // first entry of the row in hv (a multiple of 8)
  const float *__restrict__ hv;       // the rows' contributions in row order (written by the stage pass of this pull)
  uint32_t *__restrict__ sum;         // the row's sum (bit pattern)
  uint32_t n;                         // selected rows
  uint32_t n_vlong;                   // of them so long that a workgroup takes the row (the first n_vlong)
};

// STAGE: chunk c's slice of contrib_in into LDS, then hv[place] = slice[local id] for the chunk's entries
__global__ void __launch_bounds__(PR_REF_STAGE_THREADS)
pr_ref_stage_kernel(const float *__restrict__ contrib_in, uint32_t m_space, const eoff_t *__restrict__ cptr,
                    const uint16_t *__restrict__ cu16, const uint32_t *__restrict__ cdest, float *__restrict__ hv,
                    const unsigned *__restrict__ skip) {
  if (skip && *skip) return;
  extern __shared__ __attribute__((aligned(16))) float s_x[];
  const uint32_t c = blockIdx.x;
  const eoff_t e0 = cptr[c], e1 = cptr[c + 1];
  if (e1 <= e0) return;
  const uint32_t base = c << PR_REF_LOG_CHUNK;
  const uint32_t n = m_space - base < (1u << PR_REF_LOG_CHUNK) ? m_space - base : (1u << PR_REF_LOG_CHUNK);
  for (uint32_t i = threadIdx.x; i < n; i += PR_REF_STAGE_THREADS) s_x[i] = contrib_in[base + i];
  __syncthreads();
  constexpr int U = 4;
  for (eoff_t e = e0 + threadIdx.x; e < e1; e += (eoff_t)U * PR_REF_STAGE_THREADS) {
    uint32_t u[U], d[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
      const eoff_t ee = e + (eoff_t)k * PR_REF_STAGE_THREADS;
      u[k] = ee < e1 ? (uint32_t)__builtin_nontemporal_load(cu16 + ee) : 0u;
      d[k] = ee < e1 ? __builtin_nontemporal_load(cdest + ee) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int k = 0; k < U; k++)
      if (d[k] != 0xFFFFFFFFu) hv[d[k]] = s_x[u[k]];
  }
}

// the N values of a lane of block j0 of a row (bit patterns; +0 behind the row's end).  Two 16-byte loads, straight-line: the
// addresses depend on nothing that is loaded, so the caller keeps as many blocks in flight as it likes.
typedef float pr_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pr_ref_load(const pr_f32x4 *__restrict__ H4, uint32_t j0, unsigned lane, pr_f32x4 &a, pr_f32x4 &b) {
  const uint32_t j = j0 + lane * PR_REF_N;
  a = __builtin_nontemporal_load(H4 + (j >> 2));
  b = __builtin_nontemporal_load(H4 + (j >> 2) + 1);
}
__device__ __forceinline__ void pr_ref_mask(const pr_f32x4 &a, const pr_f32x4 &b, uint32_t j0, unsigned lane, uint32_t deg,
                                            uint32_t (&x)[PR_REF_N]) {
  const uint32_t j = j0 + lane * PR_REF_N;
  const float v[PR_REF_N] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int k = 0; k < PR_REF_N; k++) x[k] = j + (uint32_t)k < deg ? __float_as_uint(v[k]) : 0u;
}

// SCAN, a wave per row (the rows behind the very long ones)
__global__ void __launch_bounds__(GDN_BLOCK)
pr_refscan_kernel(PrRefRows rr, const unsigned *__restrict__ skip) {
  if (skip && *skip) return;
  const unsigned lane = gdn_lane();
  const uint64_t i = (uint64_t)rr.n_vlong + (((uint64_t)blockIdx.x * GDN_BLOCK + threadIdx.x) >> 6);
  if (i >= rr.n) return;
  const uint32_t deg = rr.deg[i];
  const pr_f32x4 *__restrict__ H4 = reinterpret_cast<const pr_f32x4 *>(rr.hv + rr.off[i]);
  uint32_t S = 0u;
  pr_f32x4 a0, b0, a1, b1, a2, b2;  // three blocks in flight (hv carries the slack behind the last row)
  pr_ref_load(H4, 0u, lane, a0, b0);
  pr_ref_load(H4, PR_REF_BLOCK, lane, a1, b1);
  for (uint32_t j0 = 0; j0 < deg; j0 += PR_REF_BLOCK) {
    pr_ref_load(H4, j0 + 2u * PR_REF_BLOCK, lane, a2, b2);
    uint32_t x[PR_REF_N];
    pr_ref_mask(a0, b0, j0, lane, deg, x);
    S = seq_block<PR_REF_N>(S, x, lane);
    a0 = a1;
    b0 = b1;
    a1 = a2;
    b1 = b2;
  }
  if (lane == 0) rr.sum[i] = S;
}

// SCAN, a workgroup per very long row.  A round = one block per wave, all waves at once: each loads its block and reduces
// it to ONE pair on the binade the running sum is in (seq_block_pair); then every wave chains the round's pairs -- a dozen
// scalar operations per block -- and holds the new running sum.  Where the chain meets the end of the binade (a few dozen
// times per row) the wave that owns the block adds it exactly (seq_block), and the waves behind it redo their pairs on the
// new binade.
#define PR_REFW_WAVES 16
#define PR_REFW_THREADS (64 * PR_REFW_WAVES)
__global__ void __launch_bounds__(PR_REFW_THREADS)
pr_refscan_wg_kernel(PrRefRows rr, const unsigned *__restrict__ skip, int dbg = 0 /* GDN_EXPERIMENTS: 1 no pairs, 2 no chain (timing only) */) {
  if (skip && *skip) return;
  __shared__ uint32_t s_a0[2][PR_REFW_WAVES], s_a1[2][PR_REFW_WAVES], s_S;
  const unsigned lane = gdn_lane(), w = threadIdx.x >> 6;
  const uint64_t i = blockIdx.x;
  const uint32_t deg = rr.deg[i];
  const pr_f32x4 *__restrict__ H4 = reinterpret_cast<const pr_f32x4 *>(rr.hv + rr.off[i]);
  uint32_t S = 0u;
  unsigned par = 0;
  const uint32_t round_len = PR_REFW_WAVES * PR_REF_BLOCK;
  pr_f32x4 a0, b0, a1, b1, a2, b2;  // this wave's blocks of rounds r, r + 1, r + 2
  pr_ref_load(H4, w * PR_REF_BLOCK, lane, a0, b0);
  pr_ref_load(H4, round_len + w * PR_REF_BLOCK, lane, a1, b1);
  for (uint32_t r0 = 0; r0 < deg; r0 += round_len) {
    const uint32_t j0 = r0 + w * PR_REF_BLOCK;
    pr_ref_load(H4, j0 + 2u * round_len, lane, a2, b2);
    uint32_t x[PR_REF_N];
    pr_ref_mask(a0, b0, j0, lane, deg, x);
    const bool mine = j0 < deg;  // (wave-uniform)
    uint32_t E = S >> 23;
    SeqPair t = {0u, 0u};
#ifdef GDN_EXPERIMENTS
    if (dbg & 1) {
      t.a0 = t.a1 = x[0] & 1023u;
    } else
#endif
    if (mine && E - 1u < 254u) t = seq_block_pair<PR_REF_N>(E, x, lane);
    if (lane == 0) {
      s_a0[par][w] = t.a0;
      s_a1[par][w] = t.a1;
    }
    __syncthreads();
#ifdef GDN_EXPERIMENTS
    if (dbg & 2) {
      S = ((S + s_a0[par][lane & 15u]) & 0x7FFFFFu) | 0x30000000u;
      S = (uint32_t)__builtin_amdgcn_readfirstlane((int)S);
      par ^= 1u;
      a0 = a1;
      b0 = b1;
      a1 = a2;
      b1 = b2;
      continue;
    }
#endif
    const unsigned nw = deg - r0 >= round_len ? (unsigned)PR_REFW_WAVES : (unsigned)((deg - r0 + PR_REF_BLOCK - 1) / PR_REF_BLOCK);  // blocks of this round
    unsigned first = 0;
    for (;;) {  // chain the round's pairs; every wave does, and ends with the same running sum
      // The chain is itself a composition of parity functions: lane ww holds wave ww's pair (the identity in front of `first`
      // and behind the round's last block), four DPP steps give every prefix, one ballot finds the first block in which the
      // prefix leaves the binade.  (Chained block after block by scalar code this was 2.2 of a round's 4 us: 16 waves x 16
      // dependent steps, sessions/r06_20.sh.)
      const bool in_chain = lane >= first && lane < nw;
      SeqPair lp;
      lp.a0 = in_chain ? s_a0[par][lane & (PR_REFW_WAVES - 1)] : 0u;
      lp.a1 = in_chain ? s_a1[par][lane & (PR_REFW_WAVES - 1)] : 0u;
      lp = seq_row_scan(lp);
      const uint32_t P0 = (S & 0x7FFFFFu) | 0x800000u;
      const uint32_t tot = P0 + ((P0 & 1u) ? lp.a1 : lp.a0);
      const bool normal = E - 1u < 254u;
      const unsigned long long cross = __ballot(in_chain && (!normal || tot >= (1u << 24)));
      unsigned wc = PR_REFW_WAVES;
      if (cross) {
        wc = (unsigned)__ffsll((long long)cross) - 1u;
        const uint32_t Pb = wc ? (uint32_t)__builtin_amdgcn_readlane((int)tot, wc - 1u) : P0;  // (lanes in front of `first` hold P0)
        if (normal) S = (E << 23) | (Pb & 0x7FFFFFu);
      } else if (nw > first && normal) {
        S = (E << 23) | ((uint32_t)__builtin_amdgcn_readlane((int)tot, nw - 1u) & 0x7FFFFFu);
      }
      if (wc == PR_REFW_WAVES) break;
      if (w == wc) {  // the end of the binade (or a running sum that is not a normal number yet) lies in my block: added exactly
        S = seq_block<PR_REF_N>(S, x, lane);
        if (lane == 0) s_S = S;
      }
      __syncthreads();
      S = s_S;
      E = S >> 23;
      first = wc + 1;
      if (w >= first) {  // the blocks behind it: their pairs again, on the binade the sum is in now
        SeqPair t2 = {0u, 0u};
        if (mine && E - 1u < 254u) t2 = seq_block_pair<PR_REF_N>(E, x, lane);
        if (lane == 0) {
          s_a0[par][w] = t2.a0;
          s_a1[par][w] = t2.a1;
        }
      }
      __syncthreads();
    }
    par ^= 1u;
    a0 = a1;
    b0 = b1;
    a1 = a2;
    b1 = b2;
  }
  if (threadIdx.x == 0) rr.sum[i] = S;
}

// the scores of the selected rows in front of the pull
__global__ void __launch_bounds__(GDN_BLOCK)
pr_ref_old_kernel(const uint32_t *__restrict__ row, uint32_t n, const float *__restrict__ scores, float *__restrict__ old) {
  for (uint64_t i = (uint64_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * GDN_BLOCK) old[i] = scores[row[i]];
}
// ... and behind it: scores / next contributions from the reference-order sums, the change of the L1 change per workgroup
__global__ void __launch_bounds__(GDN_BLOCK)
pr_ref_apply_kernel(const uint32_t *__restrict__ row, const uint32_t *__restrict__ sum, const float *__restrict__ old, uint32_t n,
                    float *__restrict__ scores, float *__restrict__ contrib_out, const int32_t *__restrict__ out_degree,
                    float base_score, float damping, double *__restrict__ partial, const unsigned *__restrict__ skip) {
  __shared__ double s_red[GDN_WAVES_PER_BLOCK];
  double acc = 0.0;
  if (!(skip && *skip)) {
    for (uint64_t i = (uint64_t)blockIdx.x * GDN_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * GDN_BLOCK) {
      const uint32_t r = row[i];
      const float new_score = gdn_fadd(base_score, gdn_fmul(damping, __uint_as_float(sum[i])));
      acc += (double)fabsf(gdn_fsub(new_score, old[i])) - (double)fabsf(gdn_fsub(scores[r], old[i]));
      scores[r] = new_score;
      contrib_out[r] = __fdiv_rn(new_score, (float)out_degree[r]);
    }
  }
  acc = gdn_block_sum(acc, s_red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ void pr_ref_adddiff_kernel(const double *__restrict__ partial, uint32_t n, double *__restrict__ diff) {
  if (threadIdx.x || blockIdx.x) return;
  double t = 0.0;
  for (uint32_t i = 0; i < n; i++) t += partial[i];
  *diff += t;
}
#define PR_REF_DIFF_BLOCKS 256

// plan build of the mode: sort keys of the rows (selected: ~degree << 32 | state row; others: all ones), ...
__global__ void __launch_bounds__(GDN_BLOCK)
pr_ref_keys_kernel(const eoff_t *__restrict__ rowptr, const uint32_t *__restrict__ row_ids, int32_t m_rows, uint32_t min_deg,
                   unsigned long long *__restrict__ keys, unsigned long long *__restrict__ count /* [0] selected rows, [1] their entries */) {
  __shared__ unsigned long long s_tmp[GDN_WAVES_PER_BLOCK];
  unsigned long long sel = 0, ent = 0;
  for (uint64_t k = (uint64_t)blockIdx.x * GDN_BLOCK + threadIdx.x; k < (uint64_t)m_rows; k += (uint64_t)gridDim.x * GDN_BLOCK) {
    const uint64_t r = row_ids ? (uint64_t)row_ids[k] : k;
    const eoff_t d = rowptr[r + 1] - rowptr[r];
    const bool s = d >= (eoff_t)min_deg && d > 0 && d < 0xFFFFFFFFull;
    keys[k] = s ? ((unsigned long long)(0xFFFFFFFFu - (uint32_t)d) << 32) | k : ~0ull;
    sel += s ? 1u : 0u;
    ent += s ? d : 0u;
  }
  gdn_block_add_u64(sel, count, s_tmp);
  gdn_block_add_u64(ent, count + 1, s_tmp);
}
// ... the sorted keys -> row, degree, padded length
__global__ void __launch_bounds__(GDN_BLOCK)
pr_ref_rows_kernel(const unsigned long long *__restrict__ keys, uint32_t n, uint32_t *__restrict__ row, uint32_t *__restrict__ deg,
                   uint32_t *__restrict__ padded) {
  const uint64_t i = (uint64_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  const uint32_t d = 0xFFFFFFFFu - (uint32_t)(k >> 32);
  row[i] = (uint32_t)k;
  deg[i] = d;
  padded[i] = (d + 7u) & ~7u;
}
// ... the entries of the rows as sort keys (chunk of the source in the plan's vertex space << 32 | place in hv); the pad places
// behind a row's end get the key of a chunk that does not exist and sort to the end.  cmap: caller's id -> state index of a
// squished plan.  A wave per row.
__global__ void __launch_bounds__(GDN_BLOCK)
pr_ref_ckeys_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const uint32_t *__restrict__ row_ids,
                    const eoff_t *__restrict__ cmap, const uint32_t *__restrict__ row, const uint32_t *__restrict__ deg,
                    const eoff_t *__restrict__ off, uint32_t n, unsigned long long *__restrict__ keys, uint32_t *__restrict__ cols) {
  const unsigned lane = gdn_lane();
  for (uint64_t i = ((uint64_t)blockIdx.x * GDN_BLOCK + threadIdx.x) >> 6; i < n; i += ((uint64_t)gridDim.x * GDN_BLOCK) >> 6) {
    const uint64_t r = row_ids ? (uint64_t)row_ids[row[i]] : (uint64_t)row[i];
    const eoff_t e0 = rowptr[r], o = off[i];
    const uint32_t d = deg[i], dp = (d + 7u) & ~7u;
    for (uint32_t j = lane; j < dp; j += 64) {
      unsigned long long key = (0xFFFFull << 32) | (unsigned long long)(o + j);
      uint32_t c = 0u;
      if (j < d) {
        const vid_t v = colidx[e0 + j];
        c = cmap ? (uint32_t)cmap[v] : (uint32_t)v;
        key = ((unsigned long long)(c >> PR_REF_LOG_CHUNK) << 32) | (unsigned long long)(o + j);
      }
      keys[o + j] = key;
      cols[o + j] = c;
    }
  }
}
// ... and, from the keys sorted by chunk: place and local source id of every entry, the first entry of every chunk that has one
__global__ void __launch_bounds__(GDN_BLOCK)
pr_ref_cfill_kernel(const unsigned long long *__restrict__ keys, uint64_t n_real, const uint32_t *__restrict__ cols,
                    uint32_t *__restrict__ cdest, uint16_t *__restrict__ cu16, eoff_t *__restrict__ cptr) {
  for (uint64_t e = (uint64_t)blockIdx.x * GDN_BLOCK + threadIdx.x; e < n_real; e += (uint64_t)gridDim.x * GDN_BLOCK) {
    const unsigned long long k = keys[e];
    const uint32_t place = (uint32_t)k, chunk = (uint32_t)(k >> 32);
    cdest[e] = place;
    cu16[e] = (uint16_t)(cols[place] & ((1u << PR_REF_LOG_CHUNK) - 1u));
    if (e == 0 || (uint32_t)(keys[e - 1] >> 32) != chunk) cptr[chunk] = e;
  }
}

// bin and in-bin index of every hub row in the compacted main layout (bin_lo = first original row of a bin,
// dst_bits = rows that have entries)
__global__ void __launch_bounds__(GDN_BLOCK)
pr_hubrow_locate_kernel(const uint32_t *__restrict__ ids, unsigned n, const uint32_t *__restrict__ bin_lo, unsigned nbins,
                        const uint32_t *__restrict__ dst_bits, unsigned *__restrict__ bin_of, uint16_t *__restrict__ vl) {
  const unsigned k = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (k >= n) return;
  const unsigned r = ids[k];
  unsigned lo = 0, hi = nbins;  // last bin with bin_lo <= r
  while (hi - lo > 1) {
    const unsigned mid = (lo + hi) >> 1;
    if (bin_lo[mid] <= r) lo = mid;
    else hi = mid;
  }
  unsigned rank = 0;  // active rows of the bin in front of r
  const unsigned b0 = bin_lo[lo];
  for (unsigned w = b0 >> 5; w <= (r >> 5); w++) {
    unsigned bits = dst_bits[w];
    if (w == (b0 >> 5)) bits &= ~0u << (b0 & 31u);
    if (w == (r >> 5)) bits &= (1u << (r & 31u)) - 1u;
    rank += (unsigned)__popc(bits);
  }
  bin_of[k] = lo;
  vl[k] = (uint16_t)rank;
}

__global__ void __launch_bounds__(GDN_BLOCK)
pr_count_sources_kernel(const int32_t *__restrict__ deg, int32_t m, unsigned long long *__restrict__ out) {
  unsigned long long n = 0;
  for (size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; v < (size_t)m; v += (size_t)gridDim.x * GDN_BLOCK) n += deg[v] > 0;
  n = gdn_wave_sum(n);
  if (gdn_lane() == 0 && n) atomicAdd(out, n);
}

// How local is the gather of a pull over this in-CSR?  One wave per 16th row counts its edges and those whose source
// lies within 2^16 ids of the row: on a lattice / banded / ring-like graph nearly all of them do, consecutive rows then
// read the same few lines of the contribution vector out of L1 / L2, and the merge-path layout beats the blocked one,
// which pays 12 B per edge whatever the structure (4096 x 4096 lattice: 0.62 against 0.53 of the roofline, small world
// 0.53 against 0.46; uniform random: 0.08 against 0.46 -- profiles/r03_shapes_layout_ab.json).
__global__ void __launch_bounds__(GDN_BLOCK)
pr_locality_sample_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, int32_t m, int32_t row_base,
                          unsigned long long *__restrict__ out /* [0] edges sampled, [1] local ones */) {
  // waves stride over the sampled rows and keep their counts; ONE pair of atomics per workgroup (a pair per sampled row
  // was 2 x 262 K atomics on one line at RMAT-22: 3 ms of a 4.5 ms layout build); at most 256 edges of a row are looked at
  __shared__ unsigned long long s_n[GDN_WAVES_PER_BLOCK], s_l[GDN_WAVES_PER_BLOCK];
  const unsigned nwaves = gridDim.x * GDN_WAVES_PER_BLOCK;
  unsigned long long n = 0, loc = 0;
  for (uint64_t wid = blockIdx.x * GDN_WAVES_PER_BLOCK + (threadIdx.x >> 6); (wid << 4) < (uint64_t)m; wid += nwaves) {
    const uint64_t row = wid << 4;
    const eoff_t b = rowptr[row];
    eoff_t e = rowptr[row + 1];
    if (e - b > 256) e = b + 256;
    const long long me = (long long)row + row_base;
    for (eoff_t k = b + gdn_lane(); k < e; k += 64) {
      const long long d = (long long)colidx[k] - me;
      n++;
      loc += (d < 65536 && d > -65536) ? 1u : 0u;
    }
  }
  n = gdn_wave_sum(n);
  loc = gdn_wave_sum(loc);
  if (gdn_lane() == 0) {
    s_n[threadIdx.x >> 6] = n;
    s_l[threadIdx.x >> 6] = loc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tn = 0, tl = 0;
    for (int w = 0; w < GDN_WAVES_PER_BLOCK; w++) {
      tn += s_n[w];
      tl += s_l[w];
    }
    if (tn) {
      atomicAdd(out, tn);
      atomicAdd(out + 1, tl);
    }
  }
}

// true: at least 85 % of the sampled edges are local (the automatic layout choice then takes the merge-path layout)
static bool pr_gather_is_local(const gdn_graph *g, int32_t row_base) {
  if (g->nnz == 0 || g->m < 16) return false;
  DevBuf<unsigned long long> acc;
  if (acc.alloc(2) != GDN_OK || hipMemset(acc.p, 0, 16) != hipSuccess) return false;
  const uint64_t sampled = ((uint64_t)g->m + 15) >> 4;
  const unsigned nb = gdn_nblocks(sampled * 64);
  hipLaunchKernelGGL(pr_locality_sample_kernel, dim3(nb > 2048u ? 2048u : nb), dim3(GDN_BLOCK), 0, 0, g->rowptr, g->colidx, g->m,
                     row_base, acc.p);
  unsigned long long h[2] = {0, 0};
  if (hipMemcpy(h, acc.p, 16, hipMemcpyDeviceToHost) != hipSuccess) return false;
  return h[0] > 0 && (double)h[1] >= 0.85 * (double)h[0];
}

__global__ void __launch_bounds__(GDN_BLOCK)
pr_contrib_kernel(const float *__restrict__ scores, const int32_t *__restrict__ out_degree, int32_t m,
                  float *__restrict__ contrib) {
  const int32_t v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < m) contrib[v] = __fdiv_rn(scores[v], (float)out_degree[v]);
}

// ---- squished vertex space
__global__ void __launch_bounds__(GDN_BLOCK)
pr_live_flags_kernel(const eoff_t *__restrict__ rowptr, const int32_t *__restrict__ deg, int32_t m, uint32_t *__restrict__ flag) {
  const size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (size_t)m) flag[v] = (rowptr[v + 1] > rowptr[v] || deg[v] > 0) ? 1u : 0u;
}

__global__ void __launch_bounds__(GDN_BLOCK)
pr_squish_vertices_kernel(const uint32_t *__restrict__ flag, const eoff_t *__restrict__ cmap, const eoff_t *__restrict__ rowptr,
                          const int32_t *__restrict__ deg, int32_t m, uint64_t nnz, uint32_t m_state, uint32_t *__restrict__ ids,
                          int32_t *__restrict__ deg_c, eoff_t *__restrict__ rowptr_c, uint32_t *__restrict__ bits) {
  const size_t w = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;  // one 32-id word per thread
  if (w * 32 >= (size_t)m) return;
  unsigned word = 0;
  for (unsigned i = 0; i < 32; i++) {
    const size_t v = w * 32 + i;
    if (v < (size_t)m && flag[v]) {
      word |= 1u << i;
      const eoff_t k = cmap[v];
      ids[k] = (uint32_t)v;
      deg_c[k] = deg[v];
      rowptr_c[k] = rowptr[v];
    }
  }
  bits[w] = word;
  if (w == 0) rowptr_c[m_state] = nnz;
}

__global__ void __launch_bounds__(GDN_BLOCK)
pr_squish_cols_kernel(const vid_t *colidx, const eoff_t *__restrict__ cmap, uint64_t nnz, vid_t *out) {  // (out may BE colidx: gdn_pr_squish_range)
  size_t e = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * GDN_BLOCK;
  for (; e < nnz; e += stride) out[e] = (vid_t)cmap[colidx[e]];
}

// gdn_pr_squish_range: liveness from the two degree vectors (no whole in-CSR at hand) ...
__global__ void __launch_bounds__(GDN_BLOCK)
pr_live_flags_deg_kernel(const int32_t *__restrict__ in_deg, const int32_t *__restrict__ out_deg, int32_t m, uint32_t *__restrict__ flag) {
  const size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (size_t)m) flag[v] = (in_deg[v] > 0 || out_deg[v] > 0) ? 1u : 0u;
}
// ... and the offsets of the live rows of [v_lo, v_hi): a dead row has no in-edge, so the live rows' offsets are the old ones
__global__ void __launch_bounds__(GDN_BLOCK)
pr_squish_range_rows_kernel(const uint32_t *__restrict__ flag, const eoff_t *__restrict__ cmap, const eoff_t *__restrict__ rowptr,
                            int32_t v_lo, int32_t n_rows, eoff_t *__restrict__ rowptr_c) {
  const size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < (size_t)n_rows && flag[(size_t)v_lo + i]) rowptr_c[cmap[(size_t)v_lo + i] - cmap[v_lo]] = rowptr[i];
  if (i == (size_t)n_rows) rowptr_c[cmap[(size_t)v_lo + i] - cmap[v_lo]] = rowptr[i];
}

__global__ void __launch_bounds__(GDN_BLOCK)
pr_gather_state_kernel(const float *__restrict__ src, const uint32_t *__restrict__ ids, uint32_t n, float *__restrict__ dst) {
  const size_t k = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (k < n) dst[k] = src[ids[k]];
}

// export: the dead vertices get the base score, the live ones their state value
__global__ void __launch_bounds__(GDN_BLOCK)
pr_export_dead_kernel(const uint32_t *__restrict__ bits, int32_t m, float base, float *__restrict__ scores) {
  const size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (size_t)m && !((bits[v >> 5] >> (v & 31u)) & 1u)) scores[v] = base;
}
__global__ void __launch_bounds__(GDN_BLOCK)
pr_export_live_kernel(const float *__restrict__ state, const uint32_t *__restrict__ ids, uint32_t n, float *__restrict__ scores) {
  const size_t k = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (k < n) scores[ids[k]] = state[k];
}

// SUM over the dead vertices of |base - scores[v]| (their L1 change in the iteration that follows an import)
__global__ void __launch_bounds__(GDN_BLOCK)
pr_dead_diff_kernel(const float *__restrict__ scores, const uint32_t *__restrict__ bits, int32_t m, float base,
                    double *__restrict__ out) {
  double acc = 0.0;
  for (size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; v < (size_t)m; v += (size_t)gridDim.x * GDN_BLOCK)
    if (!((bits[v >> 5] >> (v & 31u)) & 1u)) acc += (double)fabsf(gdn_fsub(base, scores[v]));
  acc = gdn_wave_sum(acc);
  if (gdn_lane() == 0 && acc != 0.0) atomicAdd(out, acc);
}

// ------------------------------------------------------------------------------------------
// The whole solve in ONE launch (the reference's persistent "fusion" PageRank, src/pr/fusion.cu:45-56 over the software
// global barrier include/gbar.h:24-65) for graphs whose iteration is shorter than its launches: a cooperative grid pulls
// over the in-CSR -- a lane per short row in CSR order with the reference's fp32 operations (src/pr/omp_base.cc:24-33:
// the same bits as its sequential loop for such rows), a wave per row from 64 edges on -- writes score and next
// contribution, and meets in ONE grid barrier per iteration; behind it every workgroup adds the per-workgroup L1 changes
// in the same fixed order, so all of them take the same decision (omp_base.cc:36) without the host.  The contributions
// change hands between workgroups: device-scope (sc1) stores and loads; a vertex's score stays with one lane.
// ------------------------------------------------------------------------------------------
// Graphs of at most PR_SMALL_M vertices: ONE workgroup, the contributions of both iterations in LDS (the gathers never
// leave the CU), the iteration boundary a __syncthreads() -- an iteration of a 12 K-edge graph takes a microsecond or two
// instead of the ~17 us of a grid barrier's device-scope round trips or the ~25 us of launches and a blocking read.
#define PR_SMALL_THREADS 1024
#define PR_SMALL_M 16384
#define PR_FUSED_UNR 16  // a row of up to 16 edges: one round trip for its columns, one for its gathers
__global__ void __launch_bounds__(PR_SMALL_THREADS)
pr_small_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const int32_t *__restrict__ out_degree, int32_t m,
                float *__restrict__ scores, float base_score, float damping, double epsilon, int32_t max_iter,
                double *__restrict__ trace /* max_iter */, int32_t *__restrict__ out_iter) {
  extern __shared__ float s_contrib[];  // 2 x m
  __shared__ double s_red[PR_SMALL_THREADS / 64];
  __shared__ double s_diff;
  const unsigned lane = gdn_lane(), wave = threadIdx.x >> 6;
  float *cin = s_contrib, *cout = s_contrib + m;
  for (unsigned v = threadIdx.x; v < (unsigned)m; v += PR_SMALL_THREADS) cin[v] = __fdiv_rn(scores[v], (float)out_degree[v]);
  __syncthreads();
  int32_t iter = 0;
  for (; iter < max_iter; iter++) {
    double d = 0.0;
    for (unsigned i0 = threadIdx.x - lane; i0 < (unsigned)m; i0 += PR_SMALL_THREADS) {  // the same lane owns a vertex every iteration
      const unsigned v = i0 + lane;
      eoff_t b = 0, e = 0;
      if (v < (unsigned)m) {
        b = rowptr[v];
        e = rowptr[v + 1];
      }
      float sum = 0.0f;
      const bool mine_long = e - b >= 64u;
      unsigned long long longs = __ballot(mine_long);
      while (longs) {
        const int leader = __ffsll((long long)longs) - 1;
        longs &= longs - 1ull;
        const eoff_t bb = __shfl(b, leader, 64), ee = __shfl(e, leader, 64);
        float part = 0.0f;
        for (eoff_t k = bb + lane; k < ee; k += 64) part = gdn_fadd(part, cin[colidx[k]]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part = gdn_fadd(part, __shfl_xor(part, o, 64));
        if ((int)lane == leader) sum = part;
      }
      if (!mine_long) {
        for (eoff_t k0 = b; k0 < e; k0 += PR_FUSED_UNR) {  // PR_FUSED_UNR column reads in flight, added in CSR order
          vid_t c[PR_FUSED_UNR];
#pragma unroll
          for (int r = 0; r < PR_FUSED_UNR; r++) c[r] = k0 + r < e ? colidx[k0 + r] : -1;
#pragma unroll
          for (int r = 0; r < PR_FUSED_UNR; r++)
            if (c[r] >= 0) sum = gdn_fadd(sum, cin[c[r]]);
        }
      }
      if (v < (unsigned)m) {
        const float old_score = scores[v];
        const float new_score = gdn_fadd(base_score, gdn_fmul(damping, sum));
        scores[v] = new_score;
        cout[v] = __fdiv_rn(new_score, (float)out_degree[v]);
        d += (double)fabsf(gdn_fsub(new_score, old_score));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (lane == 0) s_red[wave] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = 0.0;
      for (int w = 0; w < PR_SMALL_THREADS / 64; w++) a += s_red[w];
      s_diff = a;
      trace[iter] = a;
    }
    __syncthreads();
    float *tmp = cin;
    cin = cout;
    cout = tmp;
    if (s_diff < epsilon) break;  // omp_base.cc:36
  }
  if (threadIdx.x == 0) *out_iter = iter;
}

#define PR_FUSED_THREADS 256
__device__ __forceinline__ float pr_ld_dev(const float *p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void pr_st_dev(float *p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(PR_FUSED_THREADS)
pr_fused_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const int32_t *__restrict__ out_degree, int32_t m,
                float *scores, float *c0, float *c1, float base_score, float damping, double epsilon, int32_t max_iter,
                double *partial /* 2 x gridDim.x */, unsigned *bar /* GDN_GBAR_WORDS, zeroed by the host */,
                double *__restrict__ trace /* max_iter */, int32_t *__restrict__ out_iter) {
  __shared__ double s_red[PR_FUSED_THREADS / 64];
  __shared__ double s_diff;
  const unsigned lane = gdn_lane(), wave = threadIdx.x >> 6;
  const unsigned gt = blockIdx.x * PR_FUSED_THREADS + threadIdx.x, nt = gridDim.x * PR_FUSED_THREADS;
  for (unsigned v = gt; v < (unsigned)m; v += nt) pr_st_dev(c0 + v, __fdiv_rn(scores[v], (float)out_degree[v]));
  gdn_grid_barrier(bar, gridDim.x);
  const float *cin = c0;
  float *cout = c1;
  int32_t iter = 0;
  for (; iter < max_iter; iter++) {
    double d = 0.0;
    for (unsigned i0 = gt - lane; i0 < (unsigned)m; i0 += nt) {  // a wave takes 64 consecutive rows, one per lane
      const unsigned v = i0 + lane;
      eoff_t b = 0, e = 0;
      if (v < (unsigned)m) {
        b = rowptr[v];
        e = rowptr[v + 1];
      }
      float sum = 0.0f;
      // rows of a wave's width or more: the whole wave, lane l adds edges l, l + 64, ..., then a fixed shuffle tree
      const bool mine_long = e - b >= 64u;
      unsigned long long longs = __ballot(mine_long);
      while (longs) {
        const int leader = __ffsll((long long)longs) - 1;
        longs &= longs - 1ull;
        const eoff_t bb = __shfl(b, leader, 64), ee = __shfl(e, leader, 64);
        float part = 0.0f;
        for (eoff_t k = bb + lane; k < ee; k += 64) part = gdn_fadd(part, pr_ld_dev(cin + colidx[k]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part = gdn_fadd(part, __shfl_xor(part, o, 64));
        if ((int)lane == leader) sum = part;
      }
      if (!mine_long) {
        for (eoff_t k0 = b; k0 < e; k0 += PR_FUSED_UNR) {  // PR_FUSED_UNR gathers in flight, added in CSR order
          vid_t ci[PR_FUSED_UNR];
          float c[PR_FUSED_UNR];
#pragma unroll
          for (int r = 0; r < PR_FUSED_UNR; r++) ci[r] = k0 + r < e ? colidx[k0 + r] : -1;
#pragma unroll
          for (int r = 0; r < PR_FUSED_UNR; r++) c[r] = ci[r] >= 0 ? pr_ld_dev(cin + ci[r]) : 0.0f;
#pragma unroll
          for (int r = 0; r < PR_FUSED_UNR; r++)
            if (ci[r] >= 0) sum = gdn_fadd(sum, c[r]);
        }
      }
      if (v < (unsigned)m) {
        const float old_score = scores[v];
        const float new_score = gdn_fadd(base_score, gdn_fmul(damping, sum));
        scores[v] = new_score;
        pr_st_dev(cout + v, __fdiv_rn(new_score, (float)out_degree[v]));
        d += (double)fabsf(gdn_fsub(new_score, old_score));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (lane == 0) s_red[wave] = d;
    __syncthreads();
    double *mine = partial + (size_t)(iter & 1) * gridDim.x;
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < PR_FUSED_THREADS / 64; w++) t += s_red[w];
      __hip_atomic_store(mine + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gdn_grid_barrier(bar, gridDim.x);
    // the total, in the same order in every workgroup
    double t = 0.0;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += PR_FUSED_THREADS) t += __hip_atomic_load(mine + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0) s_red[wave] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = 0.0;
      for (int w = 0; w < PR_FUSED_THREADS / 64; w++) a += s_red[w];
      s_diff = a;
    }
    __syncthreads();
    const double diff = s_diff;
    if (blockIdx.x == 0 && threadIdx.x == 0) trace[iter] = diff;
    const float *tmp = cin;
    cin = cout;
    cout = const_cast<float *>(tmp);
    if (diff < epsilon) break;  // omp_base.cc:36
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *out_iter = iter;
}

// ---- the convergence test of gdn_pr's batched loop, on the device: one thread behind every iteration
struct PrLoopCtl {
  unsigned done;   // the L1 change fell below epsilon (src/pr/omp_base.cc:36): everything queued behind does nothing
  int32_t n_iter;  // iterations that ran
};
__global__ void pr_check_kernel(const double *__restrict__ diff, double epsilon, double first_extra, PrLoopCtl *ctl,
                                double *__restrict__ trace) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || ctl->done) return;
  const double d = *diff + (ctl->n_iter == 0 ? first_extra : 0.0);
  trace[ctl->n_iter] = d;
  ctl->n_iter++;
  if (d < epsilon) ctl->done = 1u;
}

// the per-iteration L1 changes of the calling thread's last gdn_pr / gdn_pr_multi solve (the reference prints them as it
// goes, src/pr/omp_base.cc:35; gdn_pr_last_trace hands them to the wrapper that prints)
static thread_local std::vector<double> g_pr_trace;
void gdn_pr_trace_set(const double *diff, int32_t n) { g_pr_trace.assign(diff, diff + (n > 0 ? n : 0)); }

// the fused solve of gdn_pr; *done = 0 when the device takes no cooperative launch (the caller runs the loop instead)
static int pr_solve_fused(const gdn_graph *g, const int32_t *d_deg, float *d_scores, float damping, double epsilon, int32_t max_iter,
                          gdn_stats *st, int *done) {
  *done = 0;
  const int32_t m = g->m;
  unsigned small_m = 1024;  // measured (R-MAT, 16 edges per vertex): 10 us per iteration at 2^10 vertices, the grid form wins from 2^12 on
  if (const char *e = gdn_test_option("GDN_PR_SMALL_M")) small_m = (unsigned)std::min(atoi(e), PR_SMALL_M);  // tuning / test knob
  const bool one_wg = (unsigned)m <= small_m;
  unsigned blocks = 1;
  if (one_wg) {
    if (hipFuncSetAttribute((const void *)pr_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PR_SMALL_M * (int)sizeof(float)) !=
        hipSuccess) {
      (void)hipGetLastError();
      return GDN_OK;
    }
  } else {
    int dev = 0, coop = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !coop ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pr_fused_kernel, PR_FUSED_THREADS, 0) != hipSuccess || per_cu < 1) {
      (void)hipGetLastError();
      return GDN_OK;
    }
    // one workgroup per CU at most, and no more workgroups than the rows fill
    blocks = (unsigned)std::min<uint64_t>((uint64_t)cus, ((uint64_t)m + PR_FUSED_THREADS - 1) / PR_FUSED_THREADS);
  }
  DevBuf<float> c0, c1;
  DevBuf<double> partial, trace;
  DevBuf<unsigned> bar;
  DevBuf<int32_t> out_iter;
  if (!one_wg) {
    GDN_TRY(c0.alloc((size_t)m));
    GDN_TRY(c1.alloc((size_t)m));
    GDN_TRY(partial.alloc(2 * (size_t)blocks));
    GDN_TRY(bar.alloc(GDN_GBAR_WORDS));
  }
  GDN_TRY(trace.alloc((size_t)max_iter));
  GDN_TRY(out_iter.alloc(1));
  HostTimer tsolve;
  tsolve.start();  // timed region == src/pr/base.cu:110-128
  const float base = (1.0f - damping) / (float)m;  // as gdn_pr_pull_rows_dev computes it
  if (one_wg) {
    hipLaunchKernelGGL(pr_small_kernel, dim3(1), dim3(PR_SMALL_THREADS), 2 * (size_t)m * sizeof(float), 0, g->rowptr, g->colidx, d_deg, m,
                       d_scores, base, damping, epsilon, max_iter, trace.p, out_iter.p);
    GDN_HIP(hipGetLastError());
  } else {
    GDN_HIP(hipMemsetAsync(bar.p, 0, GDN_GBAR_WORDS * sizeof(unsigned), 0));
    const eoff_t *a_rowptr = g->rowptr;
    const vid_t *a_colidx = g->colidx;
    int32_t a_m = m, a_max_iter = max_iter;
    float *a_scores = d_scores, *a_c0 = c0.p, *a_c1 = c1.p;
    float a_base = base, a_damping = damping;
    double a_eps = epsilon;
    double *a_partial = partial.p, *a_trace = trace.p;
    unsigned *a_bar = bar.p;
    int32_t *a_out = out_iter.p;
    void *args[] = {&a_rowptr, &a_colidx, &d_deg, &a_m, &a_scores, &a_c0, &a_c1, &a_base, &a_damping, &a_eps, &a_max_iter,
                    &a_partial, &a_bar, &a_trace, &a_out};
    GDN_HIP(hipLaunchCooperativeKernel((const void *)pr_fused_kernel, dim3(blocks), dim3(PR_FUSED_THREADS), args, 0, 0));
  }
  int32_t iter = 0;
  GDN_HIP(hipMemcpy(&iter, out_iter.p, sizeof(iter), hipMemcpyDeviceToHost));
  st->solve_ms = tsolve.stop_ms();
  const int32_t n_trace = iter < max_iter ? iter + 1 : max_iter;
  g_pr_trace.resize((size_t)n_trace);
  GDN_HIP(hipMemcpy(g_pr_trace.data(), trace.p, (size_t)n_trace * sizeof(double), hipMemcpyDeviceToHost));
  st->iterations = iter + 1;  // the reference prints iter+1 (omp_base.cc:39)
  st->last_error = g_pr_trace.back();
  st->edges_traversed = g->nnz * (uint64_t)n_trace;
  *done = 1;
  return GDN_OK;
}

extern "C" {

// layout of this thread's last gdn_pr solve: GDN_LAYOUT_CSR / GDN_LAYOUT_PB, -1 = none yet (the fused small-graph solve
// counts as GDN_LAYOUT_CSR: it pulls over the caller's in-CSR)
static thread_local int g_pr_last_layout = -1;
int gdn_pr_last_layout(int32_t *layout) {
  GDN_REQUIRE(layout != nullptr, "layout");
  *layout = g_pr_last_layout;
  return GDN_OK;
}

int gdn_pr_last_trace(int32_t capacity, int32_t *n, double *diff) {
  GDN_REQUIRE(n != nullptr && capacity >= 0 && (diff != nullptr || capacity == 0), "n / diff");
  *n = (int32_t)g_pr_trace.size();
  for (int32_t i = 0; i < *n && i < capacity; i++) diff[i] = g_pr_trace[(size_t)i];
  return GDN_OK;
}

// slice sizes: as large as LDS allows on big graphs; smaller graphs keep >= 2^slices_log slices per phase BEFORE the
// vertex compaction (about 40 % of them stay on an R-MAT graph).  Whole graphs take 2^9: every bin re-reads the tier
// tables and every workgroup pays its ramp-up, so fewer, larger slices win well below one workgroup per CU (measured,
// RMAT-20 / 22 / 24: 0.082 -> 0.070, 0.160 -> 0.134, 0.456 -> 0.447 ms per iteration); row shards
// keep 2^10 (their parts pipeline wants a wave of workgroups per part, and the choice could not be measured on a node)
static int pb_pick_log(int64_t n, int max_log, int slices_log) {
  int lg = 10;
  while (lg < max_log && ((int64_t)1 << (lg + slices_log)) < n) lg++;
  return lg;
}

static int pr_plan_place(gdn_pr_plan *p, int tries, double budget_ms);
static int pr_ref_build(gdn_pr_plan *p, const gdn_graph *csr, const eoff_t *cmap);
static thread_local bool g_pr_no_place = false;  // set by gdn_pr around its own plan: one solve does not pay for a search

int gdn_pr_plan_create(const gdn_graph *in_csr, const int32_t *d_out_degree, int32_t m_global,
                       int32_t row_base, int32_t layout, gdn_pr_plan **plan) {
  GDN_REQUIRE(plan != nullptr, "plan");
  *plan = nullptr;
  GDN_REQUIRE(in_csr != nullptr && d_out_degree != nullptr, "in_csr / d_out_degree");
  GDN_REQUIRE(m_global >= in_csr->m && row_base >= 0 && row_base + in_csr->m <= m_global, "row range");
  GDN_REQUIRE(layout == GDN_LAYOUT_CSR || layout == GDN_LAYOUT_PB || layout == GDN_LAYOUT_PB_SQUISHED ||
              layout == GDN_LAYOUT_AUTO, "layout");
  GDN_REQUIRE(layout != GDN_LAYOUT_PB_SQUISHED || (row_base == 0 && in_csr->m == m_global),
              "GDN_LAYOUT_PB_SQUISHED: whole graphs only (row_base 0, m_local == m_global)");
  if (layout == GDN_LAYOUT_AUTO) {
    const char *env = gdn_option("GDN_PR_LAYOUT");
    if (env && env[0] == 'c') layout = GDN_LAYOUT_CSR;
    else if (env && env[0] == 'p') layout = GDN_LAYOUT_PB;
    else layout = (in_csr->nnz >= (1ull << 22) && !pr_gather_is_local(in_csr, row_base)) ? GDN_LAYOUT_PB : GDN_LAYOUT_CSR;
  }
  gdn_pr_plan *p = new gdn_pr_plan();
  p->layout = layout;
  p->out_degree = d_out_degree;
  p->m_global = m_global;
  p->m_base = m_global;
  p->m_orig = m_global;
  p->row_base = row_base;
  p->m_local = in_csr->m;
  p->nnz = in_csr->nnz;
  int st;
  gdn_graph sq_graph;  // the relabelled in-CSR of a squished plan (arrays owned by the plan)
  // Round 4: main layout + record tiers from ONE gather pass (pb_build_tiered_run, gdn_pbtier.hpp).  GDN_PB_BUILDER=old
  // and the knobs the new builder does not serve (8-bit rows, hub rows, no compaction) take one pb_build per layout.
  bool tiered_builder = true;
  {
    const char *be = gdn_option("GDN_PB_BUILDER"), *ve = gdn_test_option("GDN_PB_V8"), *re = gdn_test_option("GDN_PB_HUB_ROWS"),
               *ce = gdn_test_option("GDN_PB_COMPACT");
    if ((be && be[0] == 'o') || (ve && ve[0] == '1') || (re && re[0] == '1') || (ce && ce[0] == '0')) tiered_builder = false;
  }
  const gdn_graph *raw_csr = in_csr;  // the caller's graph (a squished plan relabels rows and columns)
  DevBuf<eoff_t> cmap;                // squished plan: caller's id -> state index (exclusive scan of the live flags)
  if (layout == GDN_LAYOUT_PB_SQUISHED) {
    layout = p->layout = GDN_LAYOUT_PB;
    const int32_t m = in_csr->m;
    DevBuf<uint32_t> flag;
    st = flag.alloc((size_t)m);
    if (st == GDN_OK) st = cmap.alloc((size_t)m + 1);
    if (st == GDN_OK) {
      hipLaunchKernelGGL(pr_live_flags_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, in_csr->rowptr,
                         d_out_degree, m, flag.p);
      st = gdn_exclusive_scan_u32_to_u64(flag.p, cmap.p, (size_t)m, 0);
    }
    eoff_t n_live = 0;
    if (st == GDN_OK && hipMemcpy(&n_live, cmap.p + m, sizeof(eoff_t), hipMemcpyDeviceToHost) != hipSuccess) st = GDN_ERR_HIP;
    if (st == GDN_OK && n_live > 0 && n_live < (eoff_t)m) {
      const uint32_t ms = (uint32_t)n_live;
      if ((st = p->sq_ids.alloc(ms)) == GDN_OK && (st = p->sq_deg.alloc(ms)) == GDN_OK &&
          (st = p->sq_rowptr.alloc((size_t)ms + 1)) == GDN_OK && (st = p->sq_bits.alloc(((size_t)m + 31) / 32 + 1)) == GDN_OK &&
          (tiered_builder || (st = p->sq_colidx.alloc((size_t)in_csr->nnz)) == GDN_OK) && (st = p->sq_diff.alloc(1)) == GDN_OK) {
        (void)hipMemset(p->sq_diff.p, 0, sizeof(double));
        hipLaunchKernelGGL(pr_squish_vertices_kernel, dim3(gdn_nblocks(((uint64_t)m + 31) / 32)), dim3(GDN_BLOCK), 0, 0, flag.p,
                           cmap.p, in_csr->rowptr, d_out_degree, m, in_csr->nnz, ms, p->sq_ids.p, p->sq_deg.p, p->sq_rowptr.p,
                           p->sq_bits.p);
        // (the tiered builder reads the caller's column ids through cmap: the relabelled copy -- a gather pass over the
        // edges of its own, 40 ms at RMAT-27 -- is only made for pb_build)
        if (in_csr->nnz && !tiered_builder)
          hipLaunchKernelGGL(pr_squish_cols_kernel, dim3(65536), dim3(GDN_BLOCK), 0, 0, in_csr->colidx, cmap.p, in_csr->nnz,
                             p->sq_colidx.p);
        if (hipDeviceSynchronize() != hipSuccess) {
          gdn_set_error("gdn_pr_plan_create: squish kernels failed: %s", hipGetErrorString(hipGetLastError()));
          st = GDN_ERR_HIP;
        }
        sq_graph.m = (int32_t)ms;
        sq_graph.nnz = in_csr->nnz;
        sq_graph.rowptr = p->sq_rowptr.p;
        sq_graph.colidx = tiered_builder ? raw_csr->colidx : p->sq_colidx.p;  // (raw ids: only pb_build_tiered_run may read them)
        sq_graph.owned = false;
        in_csr = &sq_graph;
        d_out_degree = p->sq_deg.p;
        m_global = (int32_t)ms;
        p->squished = true;
        p->out_degree = p->sq_deg.p;
        p->m_global = m_global;
        p->m_local = m_global;
      }
    }
    if (st != GDN_OK) {
      delete p;
      return st;
    }
  }
  if (layout == GDN_LAYOUT_CSR) {
    st = mp_plan_build(p->mp, in_csr, 0);
  } else {
    // (round 6: whole graphs 2^8 -- with fewer slices than CUs filled up to exactly one round, pb_slots_per_slice --: ONE round
    // of large chunks beats two of half-size ones on mid-size graphs: LJ-like 0.542 -> 0.565 of the roofline, R-MAT-22 0.600 ->
    // 0.614, R-MAT-24 / 25 / 27 unchanged (their slices are full-size anyway), profiles/r06_pr_midsize.txt)
    int slices_log = in_csr->m == m_global ? 8 : 10;
    if (const char *e = gdn_xoption("GDN_PB_SLICES_LOG")) slices_log = atoi(e) >= 6 && atoi(e) <= 12 ? atoi(e) : slices_log;  // tuning knob
    int lc = pb_pick_log(m_global, PB_MAX_LOG_CHUNK, slices_log), lb = pb_pick_log(in_csr->m, PB_MAX_LOG_BIN, slices_log);
    if (const char *e = gdn_test_option("GDN_PB_LOG_CHUNK")) lc = atoi(e);  // tuning knobs (tools/)
    if (const char *e = gdn_xoption("GDN_PB_LOG_BIN")) lb = atoi(e);
    // vertex compaction on by default (GDN_PB_COMPACT=0 switches it off for A/B measurements)
    const char *ce = gdn_test_option("GDN_PB_COMPACT");
    // tiles padded to 32 edges = whole 128-byte lines of vals (a line shared by two tiles is written by two
    // workgroups at different times: measured 3.9 -> 3.0 ms for phase A on RMAT-27), one G entry per 32 edges
    unsigned pad = 32;
    int lg = 5;
    if (const char *e = gdn_xoption("GDN_PB_PAD")) pad = (unsigned)atoi(e);
    if (const char *e = gdn_xoption("GDN_PB_LOG_GROUP")) lg = atoi(e);
    const bool compact = !(ce && ce[0] == '0');
    // 8-bit delta-coded rows (PbPlan::v8) are OFF by default: they save 0.94 B/edge of phase B's reads but the decode
    // (3 DPP steps + unpack per quad) cost more than that on RMAT-27 (B 3.0 -> 3.5 ms); GDN_PB_V8=1 builds them
    const char *ve = gdn_test_option("GDN_PB_V8");
    const bool v_delta = pad >= 32 && ve && ve[0] == '1';
    DevBuf<uint8_t> cls;
    PbScratch scratch;  // the key buffers of the (up to four) layout builds below
    const char *he = gdn_test_option("GDN_PB_HUBS");  // 0 switches the hub tier off (A/B measurements)
    st = GDN_OK;
    uint64_t hub_min_nnz = 1ull << 24;  // below this the second layout does not pay for itself
    if (const char *e = gdn_test_option("GDN_PB_HUB_MIN_NNZ")) hub_min_nnz = strtoull(e, nullptr, 10);  // test knob
    const char *me = gdn_test_option("GDN_PB_MID");  // number of mid tiers (0 switches them off; A/B measurements)
    int max_mid = me ? atoi(me) : PB_MAX_MID;
    if (max_mid < 0 || lb > PB_MID_ROW_BITS) max_mid = 0;
    DevBuf<uint32_t> mid_ids[PB_MAX_MID];
    unsigned n_mid[PB_MAX_MID] = {};
    const bool want_tiers = compact && in_csr->nnz >= hub_min_nnz && !(he && he[0] == '0');
    bool built = false;  // by the tiered builder
    if (tiered_builder && lb <= PB_MID_ROW_BITS) {
      PbTieredArgs ta;
      PbTierSet ts;
      ta.rowptr = in_csr->rowptr;
      ta.colidx = raw_csr->colidx;
      ta.colmap = p->squished ? cmap.p : nullptr;
      ta.m_raw = raw_csr->m;
      ta.m_rows = in_csr->m;
      ta.m_global = m_global;
      ta.nnz = in_csr->nnz;
      // the out-degrees ARE the column counts when the plan covers the whole graph (a row shard sees a part of every
      // column: exact marks there); a mismatch is detected by the gather pass (rc 2) and the build repeated without
      ta.src_count = in_csr->m == m_global ? d_out_degree : nullptr;
      ta.log_chunk = lc;
      ta.log_bin = lb;
      // one 1024-thread workgroup of the accumulate phase fills a CU whatever its bin's size: the bins of a row shard
      // (2^13 rows and fewer) are spread over whole rounds of workgroups as well -- RMAT-27 / 8: 793 bins = 3.1 rounds in the
      // time of 4 (profiles/r06_shard_compute.md)
      {
        // (measured in round 6, RMAT-27 / 8: 1024 bins instead of 793 -- accumulate phase 0.39-0.46 -> 0.43-0.46 ms: no gain, the
        // rounds are uneven because the BINS are; opt-in)
        const char *be = gdn_xoption("GDN_PB_BALANCE_SHARDS");
        if (be && be[0] == '1') ta.bin_balance_log = lb;
      }
      ta.pad = pad;
      ta.log_group = lg;
      ta.tiers = want_tiers;
      ta.max_mid = max_mid;
      ta.min16 = 1u;
      {  // mid-tier record streams in lane-interleaved blocks (a quarter of phase B's record loads); GDN_PB_REC_IL=0: plain
        const char *ie = gdn_test_option("GDN_PB_REC_IL");
        ta.interleave = !(ie && ie[0] == '0');
        const char *ve = gdn_test_option("GDN_PB_V_IL");  // the same for the main stream's rows (PbPlan::v_il); 0: plain
        ta.v_interleave = !(ve && ve[0] == '0');
      }
      int rc = pb_build_tiered_run(ta, p->pb, ts);
      if (rc == 2 && !ta.colmap) {
        ta.src_count = nullptr;
        rc = pb_build_tiered_run(ta, p->pb, ts);
      }
      if (rc < 0) st = rc;
      else if (rc == GDN_OK) {
        built = true;
        int k = 0;
        if (ts.n > 0 && ts.first_is_hub) {
          p->n_hubs = ts.t[0].n_src;
          p->hub_ids.take(ts.t[0].ids);
          p->hub_rec.take(ts.t[0].rec);
          p->hub.bin_ptr.take(ts.t[0].bin_ptr);
          p->hub.nnz = ts.t[0].nnz;
          p->hub.nbins = p->pb.nbins;
          p->hub.nchunks = 1;
          st = p->hub_val.alloc(PB_HUB_SLOTS + 3);
          p->has_hub = true;
          k = 1;
        }
        for (int t = k; t < ts.n && st == GDN_OK; t++) {
          gdn_pr_plan::MidTier &mt = p->mid[t - k];
          mt.n = ts.t[t].n_src;
          mt.ids.take(ts.t[t].ids);
          mt.rec.take(ts.t[t].rec);
          mt.layout.bin_ptr.take(ts.t[t].bin_ptr);
          mt.layout.nnz = ts.t[t].nnz;
          mt.layout.nbins = p->pb.nbins;
          mt.il = ts.t[t].interleaved;
          st = mt.val.alloc((size_t)mt.n + 4);
          p->n_mid_tiers = t - k + 1;
        }
      } else if (p->squished) {
        // outside the builder's limits (or counts that do not match the columns): pb_build below needs the relabelled columns
        if ((st = p->sq_colidx.alloc((size_t)in_csr->nnz)) == GDN_OK && in_csr->nnz) {
          hipLaunchKernelGGL(pr_squish_cols_kernel, dim3(65536), dim3(GDN_BLOCK), 0, 0, raw_csr->colidx, cmap.p, in_csr->nnz,
                             p->sq_colidx.p);
          sq_graph.colidx = p->sq_colidx.p;
        }
      }
    } else if (p->squished && tiered_builder) {
      if ((st = p->sq_colidx.alloc((size_t)in_csr->nnz)) == GDN_OK && in_csr->nnz) {
        hipLaunchKernelGGL(pr_squish_cols_kernel, dim3(65536), dim3(GDN_BLOCK), 0, 0, raw_csr->colidx, cmap.p, in_csr->nnz,
                           p->sq_colidx.p);
        sq_graph.colidx = p->sq_colidx.p;
      }
    }
    if (!built && st == GDN_OK && want_tiers)
      // four mid tiers down to 1/16 edge per source and bin: RMAT-27 62 % of the edges in record tiers, 4.12 -> 3.78 ms
      // per iteration against two tiers down to 1/4 (profiles/r03_pb_tier_sweep_reps.txt; a fifth and sixth tier give it
      // back: their table lines are fetched for one record each)
      st = pb_pick_tiers(in_csr, m_global, lb, cls, p->hub_ids, &p->n_hubs, max_mid, mid_ids, n_mid, 1u);
    const bool any_class = !built && (p->n_hubs || n_mid[0]);
    // hub rows: as many as phase A's LDS can hold accumulators for behind the slice (the slice size is known when
    // this plan covers the whole graph: sources = vertices with out-edges; a row shard takes the safe bound)
    DevBuf<uint8_t> dcls;
    // OFF by default (GDN_PB_HUB_ROWS=1 builds it): it takes 1.1 GB out of an iteration (phase B -0.2 ms) but phase A
    // pays the same back -- any wave of a CU that folds instead of streaming lowers the CU's bytes in flight
    const char *re = gdn_test_option("GDN_PB_HUB_ROWS");
    const unsigned lds_static = 8704;             // s_bits + s_pref + s_scr of pb_expand_kernel, rounded up
    unsigned slots_assumed = 1u << lc;
    if (st == GDN_OK && compact && lc == PB_MAX_LOG_CHUNK && in_csr->nnz >= hub_min_nnz && re && re[0] == '1') {
      if (in_csr->m == m_global) {
        DevBuf<unsigned long long> nsrc;
        unsigned long long h_nsrc = 0;
        if ((st = nsrc.alloc(1)) == GDN_OK) {
          (void)hipMemset(nsrc.p, 0, 8);
          hipLaunchKernelGGL(pr_count_sources_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, d_out_degree, in_csr->m, nsrc.p);
          if (hipMemcpy(&h_nsrc, nsrc.p, 8, hipMemcpyDeviceToHost) != hipSuccess) st = GDN_ERR_HIP;
          if (h_nsrc > p->n_hubs) slots_assumed = (unsigned)pb_slots_per_slice(h_nsrc - p->n_hubs, lc, PB_MAX_LOG_CHUNK);
        }
      }
      const long long room = 163840ll - 4ll * ((long long)slots_assumed + 4) - (long long)lds_static;
      unsigned max_rows = room > 0 ? (unsigned)(room / 8) : 0u;
      if (max_rows > 4096u) max_rows = 4096u;
      // a hub row should collect a couple of edges per chunk on average
      const uint64_t nchunks_est = ((uint64_t)m_global >> lc) + 1;
      if (st == GDN_OK) st = pb_pick_hub_rows(in_csr, max_rows, 2 * nchunks_est, dcls, p->hr_ids, &p->n_hr);
    }
    if (st == GDN_OK && !built)
      st = pb_build(in_csr, m_global, lc, lb, p->pb, true, nullptr, nullptr, compact, false, pad, lg,
                    any_class ? cls.p : nullptr, 0, false, v_delta, p->n_hr ? dcls.p : nullptr, 0, false, false,
                    PB_MAX_LOG_BIN, &scratch);
    if (st == GDN_OK && p->n_hr &&
        4ull * (p->pb.chunk_slots + 4ull) + 8ull * p->n_hr + lds_static > 163840ull) {
      // cannot happen with out_degree == the column counts of in_csr (the slice size was derived from it)
      gdn_set_error("gdn_pr_plan_create: out_degree does not match the columns of in_csr (slice of %u sources, %u hub rows)",
                    p->pb.chunk_slots, p->n_hr);
      st = GDN_ERR_INVALID;
    }
    if (st == GDN_OK && p->n_hr) {
      // same chunks as the main layout (the source marks ignore the row class), one bin over the hub rows
      st = pb_build(in_csr, m_global, lc, 12, p->hr, false, nullptr, nullptr, true, false, 16, 4, any_class ? cls.p : nullptr, 0,
                    false, false, dcls.p, 1, /*rows_of_class_only=*/true, /*no_gaps=*/true);
      if (st == GDN_OK && (p->hr.nbins != 1 || p->hr.nchunks != p->pb.nchunks || p->hr.chunk_slots != p->pb.chunk_slots)) {
        gdn_set_error("gdn_pr_plan_create: hub-row layout does not line up with the main layout (%u bins, %u vs %u chunks)",
                      p->hr.nbins, p->hr.nchunks, p->pb.nchunks);
        st = GDN_ERR_INVALID;
      }
      if (st == GDN_OK) st = p->hr_partial.alloc((size_t)p->pb.nchunks * p->n_hr);
      if (st == GDN_OK) st = p->hr_total.alloc(p->n_hr);
      if (st == GDN_OK) st = p->hrb_ptr.alloc((size_t)p->pb.nbins + 1);
      if (st == GDN_OK) st = p->hrb_vl.alloc(p->n_hr);
      if (st == GDN_OK) {
        DevBuf<unsigned> bin_of;
        st = bin_of.alloc(p->n_hr);
        if (st == GDN_OK) {
          hipLaunchKernelGGL(pr_hubrow_locate_kernel, dim3(gdn_nblocks(p->n_hr)), dim3(GDN_BLOCK), 0, 0, p->hr_ids.p, p->n_hr,
                             p->pb.bin_lo.p, p->pb.nbins, p->pb.dst_bits.p, bin_of.p, p->hrb_vl.p);
          std::vector<unsigned> hb(p->n_hr), ptr((size_t)p->pb.nbins + 1, 0u);
          if (hipMemcpy(hb.data(), bin_of.p, (size_t)p->n_hr * 4, hipMemcpyDeviceToHost) != hipSuccess) st = GDN_ERR_HIP;
          for (unsigned k = 0; k < p->n_hr && st == GDN_OK; k++) {
            if (hb[k] >= p->pb.nbins || (k && hb[k] < hb[k - 1])) {
              gdn_set_error("gdn_pr_plan_create: hub rows are not ordered by bin");
              st = GDN_ERR_INVALID;
            } else ptr[hb[k] + 1]++;
          }
          for (unsigned b = 0; b < p->pb.nbins; b++) ptr[b + 1] += ptr[b];
          if (st == GDN_OK &&
              hipMemcpy(p->hrb_ptr.p, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice) != hipSuccess) st = GDN_ERR_HIP;
        }
      }
      if (st == GDN_OK) {
        p->hr.G.release();
        p->has_hr = true;
      }
    }
    if (st == GDN_OK && !built && p->n_hubs) {
      st = pb_build(in_csr, m_global, PB_HUB_LOG, lb, p->hub, false, nullptr, nullptr, true, false, 16, 4, cls.p, 1, true, false,
                    nullptr, 0, false, false, PB_MAX_LOG_BIN, &scratch);
      if (st == GDN_OK && (p->hub.nchunks != 1 || p->hub.nbins != p->pb.nbins)) {
        gdn_set_error("gdn_pr_plan_create: hub layout does not line up with the main layout (%u chunks, %u vs %u bins)",
                      p->hub.nchunks, p->hub.nbins, p->pb.nbins);
        st = GDN_ERR_INVALID;
      }
      if (st == GDN_OK) st = p->hub_val.alloc(PB_HUB_SLOTS + 3);  // + the window behind the last slot
      // read by phase B as one record stream like the mid tiers (U, V and G of the layout are released)
      if (st == GDN_OK) st = pb_mid_finish(p->hub, p->n_hubs, p->hub_rec);
      if (st == GDN_OK) p->has_hub = true;
    }
    for (int t = 0; t < PB_MAX_MID && st == GDN_OK && !built && n_mid[t]; t++) {
      gdn_pr_plan::MidTier &mt = p->mid[t];
      st = pb_build(in_csr, m_global, 15, lb, mt.layout, false, nullptr, nullptr, true, false, 16, 4, cls.p, 2 + t, true, false,
                    nullptr, 0, false, false, PB_MAX_LOG_BIN, &scratch);
      if (st == GDN_OK && mt.layout.nbins != p->pb.nbins) {
        gdn_set_error("gdn_pr_plan_create: mid layout %d does not line up with the main layout (%u vs %u bins)", t,
                      mt.layout.nbins, p->pb.nbins);
        st = GDN_ERR_INVALID;
      }
      if (st == GDN_OK) st = pb_mid_finish(mt.layout, n_mid[t], mt.rec);
      if (st == GDN_OK) st = mt.val.alloc((size_t)n_mid[t] + 4);
      if (st == GDN_OK) {
        mt.n = n_mid[t];
        mt.ids.take(mid_ids[t]);
        p->n_mid_tiers = t + 1;
      }
    }
    if (st == GDN_OK && (p->has_hub || p->n_mid_tiers)) {
      // launch order of phase B by all the bytes of a bin: main stream 6 B/edge, records 4 B, 16 B per row
      const eoff_t *tp[PB_MAX_REC_TIERS];
      int nt = 0;
      if (p->has_hub) tp[nt++] = p->hub.bin_ptr.p;
      for (int t = 0; t < p->n_mid_tiers; t++) tp[nt++] = p->mid[t].layout.bin_ptr.p;
      st = pb_order_bins_by_work(p->pb, nt, tp, 6.0, 4.0, 16.0);
    }
    if (st == GDN_OK && p->pb.compact) {  // row -> bin lookups of partial launches (gdn_pr_pull_rows_dev)
      p->pb.h_bin_lo.resize((size_t)p->pb.nbins + 1);
      if (hipMemcpy(p->pb.h_bin_lo.data(), p->pb.bin_lo.p, p->pb.h_bin_lo.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        gdn_set_error("gdn_pr_plan_create: bin_lo download failed");
        st = GDN_ERR_HIP;
      }
    }
    if (st == GDN_OK) {
      const int lds_a = (int)(sizeof(float) * (p->pb.chunk_slots + 4) + (p->has_hr ? 8 * p->n_hr : 0));
      const int lds_b = (int)(sizeof(unsigned long long) << p->pb.log_bin);
      hipError_t e = hipFuncSetAttribute((const void *)pb_expand_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a);
      if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)pb_expand_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a);
      if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)pb_accumulate_kernel<PrOp>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_b);
      if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)pb_accumulate_kernel<PrOp, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_b);
      if (e != hipSuccess) {
        gdn_set_error("hipFuncSetAttribute(dynamic LDS %d/%d): %s", lds_a, lds_b, hipGetErrorString(e));
        st = GDN_ERR_HIP;
      }
    }
  }
  if (st == GDN_OK && hipDeviceSynchronize() != hipSuccess) {
    gdn_set_error("gdn_pr_plan_create: layout kernels failed: %s", hipGetErrorString(hipGetLastError()));
    st = GDN_ERR_HIP;
  }
  if (st != GDN_OK) {
    delete p;
    return st;
  }
#ifdef GDN_EXPERIMENTS  // GDN_PB_UNCACHED (A/B): bit0 record streams, bit1 V, bit2 vals, bit3 U + G in uncached memory
  if (const char *e = gdn_xoption("GDN_PB_UNCACHED")) {
    const int mask = atoi(e);
    int rc2 = GDN_OK;
    if (layout != GDN_LAYOUT_CSR) {
      if (mask & 1) {
        if (p->has_hub) rc2 = p->hub_rec.rehome(hipDeviceMallocUncached);
        for (int t = 0; t < p->n_mid_tiers && rc2 == GDN_OK; t++) rc2 = p->mid[t].rec.rehome(hipDeviceMallocUncached);
      }
      if ((mask & 2) && rc2 == GDN_OK) rc2 = p->pb.V.rehome(hipDeviceMallocUncached);
      if ((mask & 4) && rc2 == GDN_OK) rc2 = p->pb.vals.rehome(hipDeviceMallocUncached);
      if ((mask & 8) && rc2 == GDN_OK) rc2 = p->pb.U.rehome(hipDeviceMallocUncached);
      if ((mask & 8) && rc2 == GDN_OK) rc2 = p->pb.G.rehome(hipDeviceMallocUncached);
    }
    if (rc2 != GDN_OK) {
      delete p;
      return rc2;
    }
  }
  // GDN_PB_VMM=<MiB>: vals (bit0 of GDN_PB_VMM_WHAT, default), U + G (bit1), V (bit2), record streams (bit3) re-homed into
  // ranges of shuffled physical chunks of that size
  if (const char *e = gdn_xoption("GDN_PB_VMM")) {
    const size_t chunk = (size_t)atoi(e) << 20;
    const char *w = gdn_xoption("GDN_PB_VMM_WHAT");
    const int what = w ? atoi(w) : 1;
    int rc2 = GDN_OK;
    if (chunk && layout != GDN_LAYOUT_CSR) {
      if (what & 1) rc2 = p->pb.vals.rehome_shuffled(chunk, 1);
      if ((what & 2) && rc2 == GDN_OK) rc2 = p->pb.U.rehome_shuffled(chunk, 2);
      if ((what & 2) && rc2 == GDN_OK) rc2 = p->pb.G.rehome_shuffled(chunk, 3);
      if ((what & 4) && rc2 == GDN_OK) rc2 = p->pb.V.rehome_shuffled(chunk, 4);
      if ((what & 8) && rc2 == GDN_OK) {
        if (p->has_hub) rc2 = p->hub_rec.rehome_shuffled(chunk, 5);
        for (int t = 0; t < p->n_mid_tiers && rc2 == GDN_OK; t++) rc2 = p->mid[t].rec.rehome_shuffled(chunk, 6 + t);
      }
    }
    if (rc2 != GDN_OK) {
      delete p;
      return rc2;
    }
  }
#endif
  if (p->squished) {  // the PB layouts hold every edge: the relabelled CSR is not read again
    p->sq_colidx.release();
    p->sq_rowptr.release();
  }
  // GDN_PR_SUM=reference: the selected rows and their column ids in the plan's vertex space (pr_ref_build; the caller's in-CSR
  // is only read here)
  if (const char *e = gdn_option("GDN_PR_SUM")) {
    if (e[0] == 'r') {
      p->ref_sum = true;
      if (const char *d = gdn_option("GDN_PR_SUM_MIN_DEGREE")) p->ref_min_deg = (uint32_t)strtoul(d, nullptr, 10);
      const int rc2 = pr_ref_build(p, raw_csr, p->squished ? cmap.p : nullptr);
      if (rc2 != GDN_OK) {
        delete p;
        return rc2;
      }
    }
  }
  // a block reserved at the start of the process (gdn_dev_reserve) becomes the plan's `vals` (scratch of one iteration: no copy)
  if (p->layout != GDN_LAYOUT_CSR && p->pb.vals.p) {
    if (void *r = gdn_reserve_take(p->pb.vals.n * sizeof(float))) {
      if (hipMemset(r, 0, p->pb.vals.n * sizeof(float)) == hipSuccess) {  // (alignment gaps between bins must read as zero)
        DevBuf<float> nb;
        nb.base = r;
        nb.p = static_cast<float *>(r);
        nb.n = p->pb.vals.n;
        p->pb.vals.swap(nb);
        if (gdn_option("GDN_PR_PLACE_TRACE")) fprintf(stderr, "[pr place] vals lives in the block reserved at process start (%p)\n", r);
      } else {
        (void)hipGetLastError();
        (void)hipFree(r);
      }
    }
  }
  // placement search (pr_plan_place).  GDN_PR_PLACE=<tries per array> (0 = off)
  // from 3 x 2^28 edges on: RMAT-26 (1.06 G edges) gains 4 % (1.93 -> 1.85 ms), RMAT-25 and RMAT-24 plans show no spread at all
  // (0.93 / 0.45 ms wherever they lie, profiles/r03_pb_placement.txt) -- it comes with allocations of several GB
  unsigned long long place_from = 3ull << 28;
  if (const char *e = gdn_test_option("GDN_PLACE_MIN_EDGES")) place_from = strtoull(e, nullptr, 10);  // (tests force the search)
  if (p->layout != GDN_LAYOUT_CSR && p->nnz >= place_from && !g_pr_no_place) {
    int tries = 3;
    if (const char *e = gdn_option("GDN_PR_PLACE")) tries = atoi(e);
    if (tries > 0) {
      double budget = 800.0;  // ms (vals: ~0.2 s for 12 candidates; the copies of phase B's streams take the rest)
      if (const char *e = gdn_xoption("GDN_PR_PLACE_BUDGET_MS")) budget = atof(e);  // (measurement sessions)
      const int rcp = pr_plan_place(p, tries, budget);
      if (rcp != GDN_OK) {
        delete p;
        return rcp;
      }
    }
  }
  *plan = p;
  return GDN_OK;
}

// ---- the same relabelling as an object of its own: a multi-GPU driver squishes the whole graph once, cuts the
// relabelled in-CSR into vertex ranges (gdn_graph_slice_rows) and builds ordinary plans on the shards
struct gdn_pr_squish {
  int32_t m_orig = 0, m_state = 0;
  gdn_graph graph;            // relabelled in-CSR (arrays below)
  DevBuf<uint32_t> ids, bits;
  DevBuf<int32_t> deg;
  DevBuf<eoff_t> rowptr;
  DevBuf<vid_t> colidx;
  DevBuf<double> diff;
};

int gdn_pr_squish_create(const gdn_graph *in_csr, const int32_t *d_out_degree, gdn_pr_squish **out) {
  GDN_REQUIRE(in_csr && d_out_degree && out, "null argument");
  *out = nullptr;
  const int32_t m = in_csr->m;
  DevBuf<uint32_t> flag;
  DevBuf<eoff_t> cmap;
  GDN_TRY(flag.alloc((size_t)m));
  GDN_TRY(cmap.alloc((size_t)m + 1));
  hipLaunchKernelGGL(pr_live_flags_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, in_csr->rowptr, d_out_degree, m,
                     flag.p);
  GDN_TRY(gdn_exclusive_scan_u32_to_u64(flag.p, cmap.p, (size_t)m, 0));
  eoff_t n_live = 0;
  GDN_HIP(hipMemcpy(&n_live, cmap.p + m, sizeof(eoff_t), hipMemcpyDeviceToHost));
  gdn_pr_squish *q = new gdn_pr_squish();
  q->m_orig = m;
  q->m_state = (int32_t)n_live;
  const uint32_t ms = (uint32_t)n_live;
  int st;
  if ((st = q->ids.alloc(ms)) == GDN_OK && (st = q->deg.alloc(ms)) == GDN_OK && (st = q->rowptr.alloc((size_t)ms + 1)) == GDN_OK &&
      (st = q->bits.alloc(((size_t)m + 31) / 32 + 1)) == GDN_OK && (st = q->colidx.alloc((size_t)in_csr->nnz)) == GDN_OK &&
      (st = q->diff.alloc(1)) == GDN_OK) {
    (void)hipMemset(q->diff.p, 0, sizeof(double));
    hipLaunchKernelGGL(pr_squish_vertices_kernel, dim3(gdn_nblocks(((uint64_t)m + 31) / 32)), dim3(GDN_BLOCK), 0, 0, flag.p, cmap.p,
                       in_csr->rowptr, d_out_degree, m, in_csr->nnz, ms, q->ids.p, q->deg.p, q->rowptr.p, q->bits.p);
    if (in_csr->nnz)
      hipLaunchKernelGGL(pr_squish_cols_kernel, dim3(65536), dim3(GDN_BLOCK), 0, 0, in_csr->colidx, cmap.p, in_csr->nnz, q->colidx.p);
    if (hipDeviceSynchronize() != hipSuccess) {
      gdn_set_error("gdn_pr_squish_create: kernels failed: %s", hipGetErrorString(hipGetLastError()));
      st = GDN_ERR_HIP;
    }
  }
  if (st != GDN_OK) {
    delete q;
    return st;
  }
  q->graph.m = q->m_state;
  q->graph.nnz = in_csr->nnz;
  q->graph.rowptr = q->rowptr.p;
  q->graph.colidx = q->colidx.p;
  q->graph.owned = false;
  *out = q;
  return GDN_OK;
}

int gdn_pr_squish_range(gdn_graph *rows, int32_t v_lo, const int32_t *d_in_degree, const int32_t *d_out_degree, int32_t m,
                        int32_t n_bounds, const int32_t *raw_bounds, int32_t *state_bounds) {
  GDN_REQUIRE(rows && d_in_degree && d_out_degree && m > 0, "null argument");
  GDN_REQUIRE(rows->owned && v_lo >= 0 && (int64_t)v_lo + rows->m <= (int64_t)m, "an owned graph of the rows [v_lo, v_lo + rows) of m");
  GDN_REQUIRE(n_bounds == 0 || (raw_bounds && state_bounds), "bounds");
  DevBuf<uint32_t> flag;
  DevBuf<eoff_t> cmap;
  GDN_TRY(flag.alloc((size_t)m));
  GDN_TRY(cmap.alloc((size_t)m + 1));
  hipLaunchKernelGGL(pr_live_flags_deg_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, d_in_degree, d_out_degree, m, flag.p);
  GDN_TRY(gdn_exclusive_scan_u32_to_u64(flag.p, cmap.p, (size_t)m, 0));
  for (int32_t r = 0; r < n_bounds; r++) {
    GDN_REQUIRE(raw_bounds[r] >= 0 && raw_bounds[r] <= m, "a bound outside [0, m]");
    eoff_t v = 0;
    GDN_HIP(hipMemcpy(&v, cmap.p + raw_bounds[r], sizeof(eoff_t), hipMemcpyDeviceToHost));
    state_bounds[r] = (int32_t)v;
  }
  eoff_t c_lo = 0, c_hi = 0;
  GDN_HIP(hipMemcpy(&c_lo, cmap.p + v_lo, sizeof(eoff_t), hipMemcpyDeviceToHost));
  GDN_HIP(hipMemcpy(&c_hi, cmap.p + v_lo + rows->m, sizeof(eoff_t), hipMemcpyDeviceToHost));
  const int32_t n_live = (int32_t)(c_hi - c_lo);
  eoff_t *rowptr_c = nullptr;
  if (gdn_plain_malloc((void **)&rowptr_c, ((size_t)n_live + 1) * sizeof(eoff_t)) != hipSuccess) {
    gdn_set_error("gdn_pr_squish_range: out of device memory");
    return GDN_ERR_OOM;
  }
  hipLaunchKernelGGL(pr_squish_range_rows_kernel, dim3(gdn_nblocks((uint64_t)rows->m + 1)), dim3(GDN_BLOCK), 0, 0, flag.p, cmap.p,
                     rows->rowptr, v_lo, rows->m, rowptr_c);
  if (rows->nnz)  // (element-wise: in place)
    hipLaunchKernelGGL(pr_squish_cols_kernel, dim3(65536), dim3(GDN_BLOCK), 0, 0, rows->colidx, cmap.p, rows->nnz, rows->colidx);
  if (hipDeviceSynchronize() != hipSuccess) {
    gdn_set_error("gdn_pr_squish_range: kernels failed: %s", hipGetErrorString(hipGetLastError()));
    (void)gdn_plain_free(rowptr_c);
    return GDN_ERR_HIP;
  }
  (void)gdn_plain_free(rows->rowptr);
  rows->rowptr = rowptr_c;
  rows->m = n_live;
  return GDN_OK;
}

int gdn_pr_squish_free(gdn_pr_squish *sq) {
  delete sq;
  return GDN_OK;
}

int gdn_pr_squish_info(const gdn_pr_squish *sq, int32_t *m_orig, int32_t *m_state, const gdn_graph **graph,
                       const int32_t **d_degrees) {
  GDN_REQUIRE(sq != nullptr, "squish");
  if (m_orig) *m_orig = sq->m_orig;
  if (m_state) *m_state = sq->m_state;
  if (graph) *graph = &sq->graph;
  if (d_degrees) *d_degrees = sq->deg.p;
  return GDN_OK;
}

int gdn_pr_squish_degrees_dev(const gdn_pr_squish *sq, int32_t *d_degrees, void *stream) {
  GDN_REQUIRE(sq && d_degrees, "null argument");
  GDN_HIP(hipMemcpyAsync(d_degrees, sq->deg.p, (size_t)sq->m_state * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return GDN_OK;
}

int gdn_pr_squish_import_dev(gdn_pr_squish *sq, const float *d_scores, float *d_state, float damping, double *dead_diff,
                             void *stream) {
  GDN_REQUIRE(sq && d_scores && d_state, "null argument");
  hipStream_t s = (hipStream_t)stream;
  const uint32_t n = (uint32_t)sq->m_state;
  hipLaunchKernelGGL(pr_gather_state_kernel, dim3(gdn_nblocks(n)), dim3(GDN_BLOCK), 0, s, d_scores, sq->ids.p, n, d_state);
  if (dead_diff) {  // blocking: the L1 change of the dead vertices in the first iteration after the import
    GDN_HIP(hipMemsetAsync(sq->diff.p, 0, sizeof(double), s));
    hipLaunchKernelGGL(pr_dead_diff_kernel, dim3(2048), dim3(GDN_BLOCK), 0, s, d_scores, sq->bits.p, sq->m_orig,
                       (1.0f - damping) / (float)sq->m_orig, sq->diff.p);
    GDN_HIP(hipStreamSynchronize(s));
    GDN_HIP(hipMemcpy(dead_diff, sq->diff.p, sizeof(double), hipMemcpyDeviceToHost));
  }
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

int gdn_pr_squish_export_dev(gdn_pr_squish *sq, const float *d_state, float *d_scores, float damping, void *stream) {
  GDN_REQUIRE(sq && d_scores && d_state, "null argument");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(pr_export_dead_kernel, dim3(gdn_nblocks((uint64_t)sq->m_orig)), dim3(GDN_BLOCK), 0, s, sq->bits.p, sq->m_orig,
                     (1.0f - damping) / (float)sq->m_orig, d_scores);
  hipLaunchKernelGGL(pr_export_live_kernel, dim3(gdn_nblocks((uint64_t)sq->m_state)), dim3(GDN_BLOCK), 0, s, d_state, sq->ids.p,
                     (uint32_t)sq->m_state, d_scores);
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

// plans built on (shards of) a squished graph: the base score is (1 - d) / m of the ORIGINAL vertex count
int gdn_pr_plan_set_base(gdn_pr_plan *plan, int32_t m_base) {
  // m_base < m_global: shards of a PADDED vertex space (gdn_graph_slice_padded: the slots hold more ids than vertices)
  GDN_REQUIRE(plan != nullptr && m_base >= 1, "m_base");
  plan->m_base = m_base;
  if (!plan->squished) plan->m_orig = m_base;  // gdn_pr_iter_bytes counts the caller's vertices
  return GDN_OK;
}

int gdn_pr_plan_state_size(const gdn_pr_plan *plan, int32_t *m_state) {
  GDN_REQUIRE(plan != nullptr && m_state != nullptr, "null argument");
  *m_state = plan->m_local;
  return GDN_OK;
}

int gdn_pr_import_dev(gdn_pr_plan *plan, const float *d_scores, float *d_state, float damping, void *stream) {
  GDN_REQUIRE(plan && d_scores && d_state, "null argument");
  hipStream_t s = (hipStream_t)stream;
  if (!plan->squished) {
    if (d_state != d_scores) GDN_HIP(hipMemcpyAsync(d_state, d_scores, (size_t)plan->m_local * 4, hipMemcpyDeviceToDevice, s));
    return GDN_OK;
  }
  const uint32_t n = (uint32_t)plan->m_local;
  hipLaunchKernelGGL(pr_gather_state_kernel, dim3(gdn_nblocks(n)), dim3(GDN_BLOCK), 0, s, d_scores, plan->sq_ids.p, n, d_state);
  GDN_HIP(hipMemsetAsync(plan->sq_diff.p, 0, sizeof(double), s));
  hipLaunchKernelGGL(pr_dead_diff_kernel, dim3(2048), dim3(GDN_BLOCK), 0, s, d_scores, plan->sq_bits.p, plan->m_orig,
                     (1.0f - damping) / (float)plan->m_base, plan->sq_diff.p);
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

int gdn_pr_import_diff(gdn_pr_plan *plan, double *dead_diff) {
  GDN_REQUIRE(plan && dead_diff, "null argument");
  *dead_diff = 0.0;
  if (plan->squished) GDN_HIP(hipMemcpy(dead_diff, plan->sq_diff.p, sizeof(double), hipMemcpyDeviceToHost));
  return GDN_OK;
}

int gdn_pr_export_dev(gdn_pr_plan *plan, const float *d_state, float *d_scores, float damping, void *stream) {
  GDN_REQUIRE(plan && d_scores && d_state, "null argument");
  hipStream_t s = (hipStream_t)stream;
  if (!plan->squished) {
    if (d_state != d_scores) GDN_HIP(hipMemcpyAsync(d_scores, d_state, (size_t)plan->m_local * 4, hipMemcpyDeviceToDevice, s));
    return GDN_OK;
  }
  const uint32_t n = (uint32_t)plan->m_local;
  hipLaunchKernelGGL(pr_export_dead_kernel, dim3(gdn_nblocks((uint64_t)plan->m_orig)), dim3(GDN_BLOCK), 0, s, plan->sq_bits.p,
                     plan->m_orig, (1.0f - damping) / (float)plan->m_base, d_scores);
  hipLaunchKernelGGL(pr_export_live_kernel, dim3(gdn_nblocks(n)), dim3(GDN_BLOCK), 0, s, d_state, plan->sq_ids.p, n, d_scores);
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

int gdn_pr_plan_free(gdn_pr_plan *plan) {
  delete plan;
  return GDN_OK;
}

int gdn_pr_contrib_dev(gdn_pr_plan *plan, const float *d_scores, float *d_contrib, void *stream) {
  GDN_REQUIRE(plan && d_scores && d_contrib, "null argument");
  hipLaunchKernelGGL(pr_contrib_kernel, dim3(gdn_nblocks((uint64_t)plan->m_local)), dim3(GDN_BLOCK), 0,
                     (hipStream_t)stream, d_scores, plan->out_degree, plan->m_local, d_contrib + plan->row_base);
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

// first bin whose first row is >= row (bins cover [0, m_local) contiguously, ascending)
static unsigned pb_first_bin_at(const PbPlan &pb, int64_t row) {
  if (row <= 0) return 0;
  if (row >= (int64_t)pb.m_local) return pb.nbins;
  if (!pb.compact) {
    const uint64_t b = ((uint64_t)row + (1ull << pb.log_bin) - 1) >> pb.log_bin;
    return b > pb.nbins ? pb.nbins : (unsigned)b;
  }
  unsigned lo = 0, hi = pb.nbins;  // smallest b with h_bin_lo[b] >= row
  while (lo < hi) {
    const unsigned mid = (lo + hi) >> 1;
    if ((int64_t)pb.h_bin_lo[mid] >= row) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}

// GDN_PR_SUM=reference, behind a pull: the selected rows summed again in the reference's order -- the stage pass (their
// contributions into hv through LDS slices), the scans (waves / workgroups, side by side on two streams) --, their scores /
// next contributions rewritten, the L1 change corrected by what that moved
static int pr_ref_resum(gdn_pr_plan *plan, const PrOp &op, double *d_diff, hipStream_t s) {
  if (plan->ref_n == 0) return GDN_OK;
  static bool attr_set = false;
  const size_t lds = sizeof(float) << PR_REF_LOG_CHUNK;
  if (!attr_set) {
    GDN_HIP(hipFuncSetAttribute((const void *)pr_ref_stage_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(pr_ref_stage_kernel, dim3(plan->ref_nchunks), dim3(PR_REF_STAGE_THREADS), lds, s, op.contrib_in, (uint32_t)plan->m_global,
                     plan->ref_cptr.p, plan->ref_cu16.p, plan->ref_cdest.p, plan->ref_hv.p, op.skip);
  PrRefRows rr;
  rr.row = plan->ref_row.p;
  rr.deg = plan->ref_deg.p;
  rr.off = plan->ref_off.p;
  rr.hv = plan->ref_hv.p;
  rr.sum = plan->ref_sumbits.p;
  rr.n = plan->ref_n;
  rr.n_vlong = plan->ref_n_vlong;
  // the very long rows' workgroups (a chain per row) run BESIDE the other rows' waves: the two kinds of rows share nothing
  hipStream_t sw = s;
  if (rr.n_vlong && plan->ref_stream && plan->ref_ev[0] && plan->ref_ev[1]) {
    sw = plan->ref_stream;
    GDN_HIP(hipEventRecord(plan->ref_ev[0], s));
    GDN_HIP(hipStreamWaitEvent(sw, plan->ref_ev[0], 0));
  }
  if (rr.n_vlong)
    hipLaunchKernelGGL(pr_refscan_wg_kernel, dim3(rr.n_vlong), dim3(PR_REFW_THREADS), 0, sw, rr, op.skip,
                       gdn_xoption("GDN_PR_REF_DBG") ? atoi(gdn_xoption("GDN_PR_REF_DBG")) : 0);
  if (rr.n > rr.n_vlong)
    hipLaunchKernelGGL(pr_refscan_kernel, dim3((rr.n - rr.n_vlong + GDN_WAVES_PER_BLOCK - 1) / GDN_WAVES_PER_BLOCK), dim3(GDN_BLOCK), 0, s, rr, op.skip);
  if (sw != s) {
    GDN_HIP(hipEventRecord(plan->ref_ev[1], sw));
    GDN_HIP(hipStreamWaitEvent(s, plan->ref_ev[1], 0));
  }
  hipLaunchKernelGGL(pr_ref_apply_kernel, dim3(PR_REF_DIFF_BLOCKS), dim3(GDN_BLOCK), 0, s, plan->ref_row.p, plan->ref_sumbits.p,
                     plan->ref_old.p, plan->ref_n, op.scores, op.contrib_out, op.out_degree, op.base_score, op.damping,
                     plan->ref_partial.p, op.skip);
  if (d_diff) hipLaunchKernelGGL(pr_ref_adddiff_kernel, dim3(1), dim3(64), 0, s, plan->ref_partial.p, (uint32_t)PR_REF_DIFF_BLOCKS, d_diff);
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

// the build of that mode (plan create): rows selected and sorted by in-degree; their entries as (chunk, place) keys, sorted by
// chunk -> the chunk-major arrays of the stage pass
static int pr_ref_build(gdn_pr_plan *p, const gdn_graph *csr, const eoff_t *cmap) {
  const int32_t m_rows = p->m_local;
  if (m_rows <= 0) return GDN_OK;
  const uint32_t *row_ids = p->squished ? p->sq_ids.p : nullptr;
  unsigned long long h[2] = {0, 0};
  const unsigned long long *sorted = nullptr;
  {
    DevBuf<unsigned long long> ka, kb, cnt;
    GDN_TRY(ka.alloc_scratch((size_t)m_rows));
    GDN_TRY(kb.alloc_scratch((size_t)m_rows));
    GDN_TRY(cnt.alloc(2));
    GDN_HIP(hipMemset(cnt.p, 0, 16));
    hipLaunchKernelGGL(pr_ref_keys_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, csr->rowptr, row_ids, m_rows, p->ref_min_deg, ka.p, cnt.p);
    GDN_HIP(hipGetLastError());
    GDN_HIP(hipMemcpy(h, cnt.p, 16, hipMemcpyDeviceToHost));
    p->ref_n = (uint32_t)h[0];
    if (p->ref_n == 0) return GDN_OK;
    GDN_TRY(gdn_radix_sort_u64(ka.p, kb.p, (unsigned long long)m_rows, 32, 64, &sorted));  // (stable: equal degrees stay in row order)
    const uint32_t n = p->ref_n;
    DevBuf<uint32_t> padded;
    GDN_TRY(p->ref_row.alloc(n));
    GDN_TRY(p->ref_deg.alloc(n));
    GDN_TRY(p->ref_off.alloc((size_t)n + 1));
    GDN_TRY(p->ref_sumbits.alloc(n));
    GDN_TRY(p->ref_old.alloc(n));
    GDN_TRY(p->ref_partial.alloc(PR_REF_DIFF_BLOCKS));
    GDN_TRY(padded.alloc_scratch(n));
    hipLaunchKernelGGL(pr_ref_rows_kernel, dim3(gdn_nblocks((uint64_t)n)), dim3(GDN_BLOCK), 0, 0, sorted, n, p->ref_row.p, p->ref_deg.p, padded.p);
    GDN_HIP(hipGetLastError());
    GDN_TRY(gdn_exclusive_scan_u32_to_u64(padded.p, p->ref_off.p, (size_t)n, 0));
    GDN_HIP(hipDeviceSynchronize());
  }
  const uint32_t n = p->ref_n;
  eoff_t total = 0;
  uint32_t longest = 0;
  GDN_HIP(hipMemcpy(&total, p->ref_off.p + n, sizeof(eoff_t), hipMemcpyDeviceToHost));
  GDN_HIP(hipMemcpy(&longest, p->ref_deg.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
  GDN_REQUIRE(total < 0xFFFF0000ull, "GDN_PR_SUM=reference: more than 2^32 entries in the selected rows (raise GDN_PR_SUM_MIN_DEGREE)");
  const uint64_t n_real = h[1];
  p->ref_edges = n_real;
  p->ref_longest = longest;
  p->ref_nchunks = (uint32_t)(((uint64_t)p->m_global + (1u << PR_REF_LOG_CHUNK) - 1) >> PR_REF_LOG_CHUNK);
  GDN_REQUIRE(p->ref_nchunks < 0xFFFFu, "GDN_PR_SUM=reference: vertex space beyond 2^31");
  // hv: the rows' places + what the scans' loads run ahead of a row's end (three rounds of a workgroup)
  GDN_TRY(p->ref_hv.alloc((size_t)total + (size_t)(3 * PR_REFW_WAVES + 4) * PR_REF_BLOCK + 8));
  GDN_HIP(hipMemset(p->ref_hv.p, 0, p->ref_hv.n * sizeof(float)));
  GDN_TRY(p->ref_cdest.alloc((size_t)n_real + 1));
  GDN_TRY(p->ref_cu16.alloc((size_t)n_real + 1));
  GDN_TRY(p->ref_cptr.alloc((size_t)p->ref_nchunks + 1));
  {
    DevBuf<unsigned long long> ka, kb;
    DevBuf<uint32_t> cols;
    GDN_TRY(ka.alloc_scratch((size_t)total + 1));
    GDN_TRY(kb.alloc_scratch((size_t)total + 1));
    GDN_TRY(cols.alloc_scratch((size_t)total + 1));
    hipLaunchKernelGGL(pr_ref_ckeys_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, csr->rowptr, csr->colidx, row_ids, cmap, p->ref_row.p, p->ref_deg.p,
                       p->ref_off.p, n, ka.p, cols.p);
    GDN_HIP(hipGetLastError());
    GDN_TRY(gdn_radix_sort_u64(ka.p, kb.p, (unsigned long long)total, 32, 48, &sorted));  // (stable: places ascend inside a chunk)
    GDN_HIP(hipMemset(p->ref_cptr.p, 0xFF, ((size_t)p->ref_nchunks + 1) * sizeof(eoff_t)));
    hipLaunchKernelGGL(pr_ref_cfill_kernel, dim3(4096), dim3(GDN_BLOCK), 0, 0, sorted, n_real, cols.p, p->ref_cdest.p, p->ref_cu16.p, p->ref_cptr.p);
    GDN_HIP(hipGetLastError());
    GDN_HIP(hipDeviceSynchronize());
  }
  {  // chunks without an entry start where the next one starts
    std::vector<eoff_t> cp((size_t)p->ref_nchunks + 1);
    GDN_HIP(hipMemcpy(cp.data(), p->ref_cptr.p, cp.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
    cp[p->ref_nchunks] = n_real;
    for (size_t c = p->ref_nchunks; c-- > 0;)
      if (cp[c] == ~(eoff_t)0) cp[c] = cp[c + 1];
    GDN_HIP(hipMemcpy(p->ref_cptr.p, cp.data(), cp.size() * sizeof(eoff_t), hipMemcpyHostToDevice));
  }
  {  // the very long rows: four rounds of a workgroup or more (the rows are sorted by length: a prefix), at most one per CU
    uint64_t thr = 4ull * PR_REFW_WAVES * PR_REF_BLOCK;
    if (const char *e = gdn_test_option("GDN_PR_SUM_WG_MIN")) thr = strtoull(e, nullptr, 10);  // (test hook: small graphs reach the workgroup kernel)
    const uint32_t look = n < 65536u ? n : 65536u;
    std::vector<uint32_t> hd(look);
    GDN_HIP(hipMemcpy(hd.data(), p->ref_deg.p, (size_t)look * 4, hipMemcpyDeviceToHost));
    uint32_t nv = 0, cap = 256u;
    if (gdn_test_option("GDN_PR_SUM_WG_MIN")) cap = look;
    while (nv < look && nv < cap && (uint64_t)hd[nv] >= thr) nv++;
    p->ref_n_vlong = nv;
    // (highest priority: the few hundred workgroups of the very long rows are a latency chain each and must not queue behind
    // the 20 K waves of the other rows, which are launched beside them and fill every CU's wave slots)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (nv && (hipStreamCreateWithPriority(&p->ref_stream, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
               hipEventCreateWithFlags(&p->ref_ev[0], hipEventDisableTiming) != hipSuccess ||
               hipEventCreateWithFlags(&p->ref_ev[1], hipEventDisableTiming) != hipSuccess)) {
      (void)hipGetLastError();  // (no second stream: everything on the caller's)
      if (p->ref_stream) (void)hipStreamDestroy(p->ref_stream);
      p->ref_stream = nullptr;
    }
  }
  if (gdn_xoption("GDN_PR_SUM_TRACE"))
    fprintf(stderr, "[pr refsum] %u rows of >= %u in-edges (%u of them on a workgroup each), %llu entries in %u chunks, longest %u\n", p->ref_n,
            p->ref_min_deg, p->ref_n_vlong, (unsigned long long)n_real, p->ref_nchunks, longest);
  return GDN_OK;
}

// parts != nullptr (gdn_pr_pull_parts_dev): the whole iteration, the accumulate launch in plan->parts_order with tickets
static int pr_pull_impl(gdn_pr_plan *plan, const float *d_contrib_in, float *d_scores, float *d_contrib_out,
                        double *d_diff, float damping, int32_t row_begin, int32_t row_end, int32_t flags,
                        void *stream, const PbParts *parts) {
  GDN_REQUIRE(plan && d_contrib_in && d_scores && d_contrib_out, "null argument");
  GDN_REQUIRE(d_contrib_in != d_contrib_out, "contrib_in and contrib_out must differ (Jacobi)");
  GDN_REQUIRE(row_begin >= 0 && row_begin <= row_end, "row range");
  const bool first = (flags & GDN_PR_PART_FIRST) != 0, last = (flags & GDN_PR_PART_LAST) != 0;
  PrOp op;
  op.contrib_in = d_contrib_in;
  op.scores = d_scores;
  op.contrib_out = d_contrib_out + plan->row_base;
  op.out_degree = plan->out_degree;
  op.base_score = (1.0f - damping) / (float)plan->m_base;
  op.damping = damping;
  op.skip = plan->skip_flag;
  op.vec_ok = ((reinterpret_cast<uintptr_t>(op.scores) | reinterpret_cast<uintptr_t>(op.contrib_out) |
                reinterpret_cast<uintptr_t>(op.out_degree)) & 15u) == 0;
  op.wt = (parts && parts->mode == 1u) ? 1 : 0;
  if (plan->ref_sum) {
    GDN_REQUIRE(first && last, "GDN_PR_SUM=reference: whole-iteration pulls only (gdn_pr_pull_dev)");
    if (plan->ref_n)
      hipLaunchKernelGGL(pr_ref_old_kernel, dim3(gdn_nblocks((uint64_t)plan->ref_n) < 4096u ? gdn_nblocks((uint64_t)plan->ref_n) : 4096u),
                         dim3(GDN_BLOCK), 0, (hipStream_t)stream, plan->ref_row.p, plan->ref_n, d_scores, plan->ref_old.p);
  }
  if (plan->layout == GDN_LAYOUT_CSR) {
    // the merge-path pass is not cut into parts: the FIRST part runs all rows, later parts are no-ops
    if (!first) return GDN_OK;
    GDN_TRY(mp_run(plan->mp, op, d_diff, (hipStream_t)stream));
    if (parts) {  // every part is ready when the one kernel is: its tickets are added behind it
      hipLaunchKernelGGL(pb_ticket_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, plan->tickets.p, *parts);
      GDN_HIP(hipGetLastError());
    }
    return plan->ref_sum ? pr_ref_resum(plan, op, d_diff, (hipStream_t)stream) : GDN_OK;
  }
  // ---- propagation-blocked path: expand (per chunk, first part) then accumulate + fused update (per bin)
  PbPlan &pb = plan->pb;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds_a = sizeof(float) * (pb.chunk_slots + 4) + (plan->has_hr ? 8 * (size_t)plan->n_hr : 0);
  const size_t lds_b = sizeof(unsigned long long) << pb.log_bin;
  const bool timed = pb.timing && pb.ev_used + 3 <= pb.ev.size();
  if (first) {
    if (timed) GDN_HIP(hipEventRecord(pb.ev[pb.ev_used], s));
    static unsigned split = 0;
    if (split == 0) {
      const char *e = gdn_xoption("GDN_PB_SPLIT");
      split = e ? (unsigned)atoi(e) : 1u;  // measured on RMAT-27: 1 -> 4.14 ms, 2 -> 4.23, 4 -> 4.53 (slice reload)
      if (split < 1 || split > 64) split = 1;
    }
    PbTierRefresh tr = PbTierRefresh();
    tr.skip = plan->skip_flag;
    if (plan->has_hub) {
      tr.ids[tr.ntiers] = plan->hub_ids.p;
      tr.val[tr.ntiers] = plan->hub_val.p;
      tr.n[tr.ntiers] = plan->n_hubs;
      tr.slots[tr.ntiers++] = (unsigned)PB_HUB_SLOTS + 3u;
    }
    for (int t = 0; t < plan->n_mid_tiers; t++) {
      tr.ids[tr.ntiers] = plan->mid[t].ids.p;
      tr.val[tr.ntiers] = plan->mid[t].val.p;
      tr.n[tr.ntiers] = plan->mid[t].n;
      tr.slots[tr.ntiers++] = plan->mid[t].n + 4u;
    }
    auto *const kern_a = plan->placing ? &pb_expand_kernel<1> : &pb_expand_kernel<0>;
    hipLaunchKernelGGL(kern_a, dim3(pb.nchunks * split), dim3(PB_THREADS), lds_a, s, d_contrib_in, pb.m_global,
                       pb.log_chunk, pb.chunk_ptr.p, pb.chunk_order.p, pb.U.p, pb.G.p, pb.vals.p,
                       pb.compact ? pb.src_bits.p : nullptr, pb.compact ? pb.chunk_lo.p : nullptr, split, pb.log_group,
#ifdef GDN_EXPERIMENTS  // GDN_PB_AVAR: A/B knobs (bit0 non-temporal stores, bit1 scalar slice loader), same results
                       gdn_xoption("GDN_PB_AVAR") ? atoi(gdn_xoption("GDN_PB_AVAR")) : 0,
#else
                       0,
#endif
                       plan->has_hr ? plan->hr.chunk_ptr.p : nullptr, plan->has_hr ? plan->hr.U.p : nullptr,
                       plan->has_hr ? plan->hr.V.p : nullptr, plan->has_hr ? plan->n_hr : 0u,
                       plan->has_hr ? plan->hr_partial.p : nullptr, pb.errflag.p, pb.chunk_slots, tr);
    if (plan->has_hr)
      hipLaunchKernelGGL(pb_hubrow_reduce_kernel, dim3((plan->n_hr + 63u) / 64u), dim3(PB_THREADS), 0, s, plan->hr_partial.p,
                         pb.nchunks, plan->n_hr, plan->hr_total.p);
    if (timed) GDN_HIP(hipEventRecord(pb.ev[pb.ev_used + 1], s));
  }
  PbMidArgs mid = PbMidArgs();
  if (plan->has_hub) {
    mid.ptr[mid.n] = plan->hub.bin_ptr.p;
    mid.rec[mid.n] = plan->hub_rec.p;
    mid.val[mid.n] = plan->hub_val.p;
    mid.zrec[mid.n] = plan->n_hubs << PB_MID_ROW_BITS;
    mid.A[mid.n] = nullptr;
    mid.form[mid.n++] = 1;
  }
  for (int t = 0; t < plan->n_mid_tiers; t++) {
    mid.ptr[mid.n] = plan->mid[t].layout.bin_ptr.p;
    mid.rec[mid.n] = plan->mid[t].rec.p;
    mid.val[mid.n] = plan->mid[t].val.p;
    mid.zrec[mid.n] = plan->mid[t].n << PB_MID_ROW_BITS;
    mid.A[mid.n] = nullptr;
    mid.form[mid.n++] = plan->mid[t].il ? 2 : 0;
  }
  mid.v_il = pb.v_il ? 1 : 0;
#ifdef GDN_EXPERIMENTS  // GDN_PB_MIDVAR: bit t = form of record tier t (A/B measurements; same results)
  if (const char *e = gdn_xoption("GDN_PB_MIDVAR"))
    for (int t = 0; t < mid.n; t++)
      if (mid.form[t] != 2) mid.form[t] = (atoi(e) >> t) & 1;  // (an interleaved stream can only be read as form 2)
#endif
  // a bin belongs to the part that holds its FIRST row: after part j every row below its row_end is final
  const bool whole = first && last;
  const unsigned b0 = whole ? 0u : pb_first_bin_at(pb, row_begin);
  const unsigned b1 = whole ? pb.nbins : (last ? pb.nbins : pb_first_bin_at(pb, row_end));
  auto *const kern_b = plan->placing ? &pb_accumulate_kernel<PrOp, 1> : &pb_accumulate_kernel<PrOp, 0>;
  if (b1 > b0)
    hipLaunchKernelGGL(kern_b, dim3(b1 - b0), dim3(PB_THREADS), lds_b, s, pb.m_local,
                       pb.log_bin, pb.bin_ptr.p, parts ? plan->parts_order.p : (whole ? pb.bin_order.p : nullptr), pb.V.p, pb.vals.p, pb.partial.p,
                       pb.errflag.p, pb.compact ? pb.dst_bits.p : nullptr, pb.compact ? pb.bin_lo.p : nullptr, op,
#ifdef GDN_EXPERIMENTS  // GDN_PB_DBG: bit0 no LDS atomics, bit1 no epilogue (TIMING ONLY, wrong results), bit2 scalar epilogue
                       gdn_xoption("GDN_PB_DBG") ? atoi(gdn_xoption("GDN_PB_DBG")) : 0,
#else
                       0,
#endif
                       b0, nullptr, nullptr, nullptr, nullptr,
                       pb.v8 ? pb.Vd.p : nullptr, pb.v8 ? pb.Vb.p : nullptr, nullptr,
                       plan->has_hr ? plan->hrb_ptr.p : nullptr, plan->has_hr ? plan->hrb_vl.p : nullptr,
                       plan->has_hr ? plan->hr_total.p : nullptr, mid, parts ? plan->tickets.p : nullptr,
                       parts ? *parts : PbParts());
  if (last) {
    if (timed) {
      GDN_HIP(hipEventRecord(pb.ev[pb.ev_used + 2], s));
      pb.ev_used += 3;
    }
    if (d_diff) {
      uint32_t n = pb.nbins;
      const double *in = pb.partial.p;
      double *bufs[2] = {pb.red_scratch.p, pb.red_scratch.p + (pb.red_scratch.n / 2)};
      int which = 0;
      for (;;) {
        const uint32_t nb = (n + MP_RED_CHUNK - 1) / MP_RED_CHUNK;
        double *out = (nb == 1) ? d_diff : bufs[which];
        hipLaunchKernelGGL(mp_reduce_f64, dim3(nb), dim3(GDN_BLOCK), 0, s, in, n, out);
        if (nb == 1) break;
        in = out;
        n = nb;
        which ^= 1;
      }
    }
  }
  GDN_HIP(hipGetLastError());
  if (plan->ref_sum) return pr_ref_resum(plan, op, d_diff, s);
  return GDN_OK;
}

int gdn_pr_pull_rows_dev(gdn_pr_plan *plan, const float *d_contrib_in, float *d_scores, float *d_contrib_out,
                         double *d_diff, float damping, int32_t row_begin, int32_t row_end, int32_t flags,
                         void *stream) {
  return pr_pull_impl(plan, d_contrib_in, d_scores, d_contrib_out, d_diff, damping, row_begin, row_end, flags, stream, nullptr);
}

// One iteration whose rows become final PART BY PART inside one launch per phase (gdn_pb.hpp, PbParts): part j = the local
// rows below row_end[j] that are not in an earlier part (a bin belongs to the part that holds its first row, as in
// gdn_pr_pull_rows_dev).  gdn_pr_wait_part_dev queues, on any OTHER stream, a one-wave kernel that ends when part j's rows of
// the LAST pull queued here are final and visible -- work queued behind it there (the all-gather of those rows) overlaps the
// accumulation of the later parts.
int gdn_pr_pull_parts_dev(gdn_pr_plan *plan, const float *d_contrib_in, float *d_scores, float *d_contrib_out,
                          double *d_diff, float damping, int32_t n_parts, const int32_t *row_end, void *stream) {
  GDN_REQUIRE(plan && row_end && n_parts >= 1 && n_parts <= PB_MAX_PARTS, "parts");
  GDN_REQUIRE(!plan->skip_flag && !plan->ref_sum, "ticketed pulls: not inside gdn_pr's batched loop / GDN_PR_SUM=reference");
  for (int j = 0; j < n_parts; j++) GDN_REQUIRE(row_end[j] >= (j ? row_end[j - 1] : 0), "row_end must ascend");
  if (!plan->tickets.p) {
    GDN_TRY(plan->tickets.alloc((size_t)PB_MAX_PARTS * PB_TICKET_STRIDE + PB_TICKET_STRIDE));
    GDN_HIP(hipMemset(plan->tickets.p, 0, plan->tickets.n * sizeof(unsigned)));
  }
  PbParts inc;  // what this pull adds to every counter
  inc.n = (unsigned)n_parts;
  if (plan->layout == GDN_LAYOUT_CSR) {
    for (int j = 0; j < n_parts; j++) inc.end[j] = 1u;
  } else {
    PbPlan &pb = plan->pb;
    const std::vector<int32_t> key(row_end, row_end + n_parts);
    if (key != plan->parts_key || !plan->parts_order.p) {
      std::vector<eoff_t> bp((size_t)pb.nbins + 1);
      GDN_HIP(hipMemcpy(bp.data(), pb.bin_ptr.p, bp.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
      std::vector<uint32_t> order(pb.nbins);
      unsigned b0 = 0, k = 0;
      for (int j = 0; j < n_parts; j++) {
        const unsigned b1 = j == n_parts - 1 ? pb.nbins : pb_first_bin_at(pb, row_end[j]);
        const unsigned first = k;
        for (unsigned b = b0; b < b1; b++) order[k++] = b;
        std::stable_sort(order.begin() + first, order.begin() + k,
                         [&](uint32_t x, uint32_t y) { return bp[x + 1] - bp[x] > bp[y + 1] - bp[y]; });
        plan->parts_launch.end[j] = k;
        if (b1 > b0) b0 = b1;
      }
      plan->parts_launch.n = (unsigned)n_parts;
      if (!plan->parts_order.p) GDN_TRY(plan->parts_order.alloc(pb.nbins ? pb.nbins : 1));
      GDN_HIP(hipStreamSynchronize((hipStream_t)stream));  // (a launch that still reads the old order)
      GDN_HIP(hipMemcpy(plan->parts_order.p, order.data(), order.size() * 4, hipMemcpyHostToDevice));
      plan->parts_key = key;
    }
    for (int j = 0; j < n_parts; j++) inc.end[j] = plan->parts_launch.end[j] - (j ? plan->parts_launch.end[j - 1] : 0u);
  }
  PbParts launch = plan->layout == GDN_LAYOUT_CSR ? inc : plan->parts_launch;
  {  // GDN_PR_TICKET_MODE: fence = an L2 write-back per workgroup, wt = write-through stores of the next contributions (default)
    const char *e = gdn_xoption("GDN_PR_TICKET_MODE");
    launch.mode = (e && e[0] == 'f') ? 0u : 1u;
  }
  GDN_TRY(pr_pull_impl(plan, d_contrib_in, d_scores, d_contrib_out, d_diff, damping, 0, plan->m_local,
                       GDN_PR_PART_FIRST | GDN_PR_PART_LAST, stream, &launch));
  for (int j = 0; j < n_parts; j++) plan->ticket_target[j] += inc.end[j];
  plan->ticket_parts = n_parts;
  return GDN_OK;
}

int gdn_pr_wait_part_dev(gdn_pr_plan *plan, int32_t part, void *stream) {
  GDN_REQUIRE(plan && plan->tickets.p && part >= 0 && part < plan->ticket_parts, "no ticketed pull with such a part was queued");
  hipLaunchKernelGGL(pb_ticket_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, plan->tickets.p + (size_t)PB_TICKET_STRIDE * part,
                     plan->ticket_target[part], plan->tickets.p + (size_t)PB_TICKET_STRIDE * PB_MAX_PARTS);
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

int gdn_pr_pull_dev(gdn_pr_plan *plan, const float *d_contrib_in, float *d_scores, float *d_contrib_out,
                    double *d_diff, float damping, void *stream) {
  GDN_REQUIRE(plan != nullptr, "plan");
  return gdn_pr_pull_rows_dev(plan, d_contrib_in, d_scores, d_contrib_out, d_diff, damping, 0, plan->m_local,
                              GDN_PR_PART_FIRST | GDN_PR_PART_LAST, stream);
}

int gdn_pr_plan_kernel_time(gdn_pr_plan *plan, int32_t reset, int32_t max_launches, double *total_ms,
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

// Placement: the arrays named in `what` (1 vals, 2 U, 4 G, 8 V, 16 hub records, 32 mid-tier records,
// 64 per-iteration tables) move into fresh allocations.
int gdn_pr_plan_move(gdn_pr_plan *plan, uint32_t what) {
  GDN_REQUIRE(plan != nullptr, "plan");
  if (plan->layout == GDN_LAYOUT_CSR) return GDN_OK;
  GDN_HIP(hipDeviceSynchronize());
  struct Tr {  // GDN_PR_PLACE_TRACE: where the arrays live after the move
    gdn_pr_plan *p;
    ~Tr() {
      if (!gdn_option("GDN_PR_PLACE_TRACE")) return;
      fprintf(stderr, "[pr move] vals %p + %zu  U %p + %zu  G %p + %zu  V %p + %zu", (void *)p->pb.vals.p, p->pb.vals.n * sizeof(*p->pb.vals.p),
              (void *)p->pb.U.p, p->pb.U.n * sizeof(*p->pb.U.p), (void *)p->pb.G.p, p->pb.G.n * sizeof(*p->pb.G.p), (void *)p->pb.V.p,
              p->pb.V.n * sizeof(*p->pb.V.p));
      if (p->has_hub) fprintf(stderr, "  hub %p + %zu", (void *)p->hub_rec.p, p->hub_rec.n * 4);
      for (int t = 0; t < p->n_mid_tiers; t++) fprintf(stderr, "  mid%d %p + %zu", t, (void *)p->mid[t].rec.p, p->mid[t].rec.n * 4);
      fprintf(stderr, "\n");
    }
  } tr{plan};
  if (what & 1u) GDN_TRY(plan->pb.vals.move());
  if (what & 2u) GDN_TRY(plan->pb.U.move());
  if (what & 4u) GDN_TRY(plan->pb.G.move());
  if (what & 8u) GDN_TRY(plan->pb.V.move());
  if ((what & 16u) && plan->has_hub) GDN_TRY(plan->hub_rec.move());
  if (what & 32u)
    for (int t = 0; t < plan->n_mid_tiers; t++) GDN_TRY(plan->mid[t].rec.move());
  if (what & 64u) {
    if (plan->has_hub) GDN_TRY(plan->hub_val.move());
    for (int t = 0; t < plan->n_mid_tiers; t++) GDN_TRY(plan->mid[t].val.move());
  }
  return GDN_OK;
}

// Placement search (PbPlacer, gdn_pb.hpp) of a blocked PageRank plan: three iterations on scratch vectors per candidate.
static int pr_plan_place(gdn_pr_plan *p, int tries, double budget_ms) {
  const int32_t ms = p->squished ? (int32_t)p->sq_ids.n : p->m_local;
  const int32_t mg = p->squished ? ms : p->m_global;
  if (ms <= 0 || mg <= 0) return GDN_OK;
  DevBuf<float> sc, c0, c1;
  DevBuf<double> diff;
  GDN_TRY(sc.alloc((size_t)ms));
  GDN_TRY(c0.alloc((size_t)mg));
  GDN_TRY(c1.alloc((size_t)mg));
  GDN_TRY(diff.alloc(1));
  const float v = 1.0f / (float)mg, c = v / 16.0f;
  GDN_TRY(gdn_fill_i32(reinterpret_cast<int32_t *>(sc.p), __builtin_bit_cast(int32_t, v), (size_t)ms, 0));
  GDN_TRY(gdn_fill_i32(reinterpret_cast<int32_t *>(c0.p), __builtin_bit_cast(int32_t, c), (size_t)mg, 0));
  GDN_TRY(gdn_fill_i32(reinterpret_cast<int32_t *>(c1.p), __builtin_bit_cast(int32_t, c), (size_t)mg, 0));
  HostTimer t;
  int it = 0;
  auto pull = [&]() {
    float *in = (it & 1) ? c1.p : c0.p, *out = (it & 1) ? c0.p : c1.p;
    it++;
    return gdn_pr_pull_dev(p, in, sc.p, out, diff.p, 0.85f, nullptr);
  };
  // Round 5: what a candidate is timed on is the PHASE that streams the array (HIP events around expand / accumulate, as
  // bench.py reads them), not the whole iteration -- phase A follows vals (1.00 <-> 1.18 ms, a fast block is about one
  // hipMalloc in six, profiles/r05_pb_channels.md), phase B the record streams and V --, and the order is by what an array
  // can gain: vals first.  vals is scratch of one iteration, so its candidates are fresh allocations without a copy
  // (~4 ms each with two timed iterations; round 4 searched it LAST, behind 16 copies of record streams, and the 1.2 s
  // budget was usually spent before it came up).
  int phase = 2;  // 0 expand, 1 accumulate, 2 both
  PbPlacer pl;
  pl.tries = tries;
  pl.budget_ms = budget_ms;
  pl.tag = "pr";
  pl.trace = gdn_option("GDN_PR_PLACE_TRACE") != nullptr;
  pl.timed = [&](double *out_ms) -> int {
    GDN_TRY(pull());
    const int n = 2;
    GDN_TRY(gdn_pr_plan_kernel_time(p, 1, n, nullptr, nullptr));
    for (int k = 0; k < n; k++) GDN_TRY(pull());
    double tot[2] = {0, 0};
    int32_t launches = 0;
    GDN_TRY(gdn_pr_plan_kernel_time(p, 0, 0, tot, &launches));
    if (launches < 1) launches = 1;
    *out_ms = (phase == 0 ? tot[0] : phase == 1 ? tot[1] : tot[0] + tot[1]) / launches;
    if (pl.trace && gdn_xoption("GDN_PR_PLACE_TRACE_AB"))  // (both phases of every timed placement, whatever it is judged on)
      fprintf(stderr, "[pr place]   phase A %.3f ms, phase B %.3f ms\n", tot[0] / launches, tot[1] / launches);
    return GDN_OK;
  };
  p->placing = true;
  int rc = pl.begin();
  int vals_tries = 4 * tries;  // GDN_PR_PLACE_VALS=<candidates>
  if (const char *e = gdn_xoption("GDN_PR_PLACE_VALS")) vals_tries = atoi(e);
  phase = 0;
  if (rc == GDN_OK) rc = pl.rebase();
  // (fast blocks run phase A at 1.00-1.01 ms, the others at 1.06-1.18: the search stops at the first candidate 10 % below the
  // slowest placement seen -- on average after six, each a 3.5 GB hipMalloc that waits for the driver to wipe pages whenever
  // another process has just freed that much: 2.0 s for twelve in session r05_06, 0.2 s on an idle box)
  // (0.90, not 0.93: the classes are ~1.14 / 1.06 / 0.99 ms -- the middle one is 7 % below the slowest, only the fast one
  // should end the search early, profiles/r05_pb_place_offsets.txt)
  double stop_ratio = 0.90;  // GDN_PR_PLACE_STOP=0: every candidate is timed (measurement sessions)
  if (const char *e = gdn_xoption("GDN_PR_PLACE_STOP")) stop_ratio = atof(e);
  if (rc == GDN_OK) rc = pl.search_fresh(p->pb.vals, "vals", vals_tries, 8, stop_ratio);
  // The arrays phase B streams (and U) are searched only on request (GDN_PR_PLACE_COPIES=1; every candidate is a copy, ~40 ms
  // per GB-sized array): timed per phase they gain nothing worth 0.3 s of plan build -- 2.536 -> 2.524 ms and 2.611 -> 2.600 ms
  // of phase B over 18 copies each in sessions r05_03 / r05_04, U 1.001 -> 0.994 ms (profiles/r05_pb_place_search.txt).
  // Round 3's "phase B follows V and the record streams by 0.03-0.1 ms each" was measured on whole iterations.
  const char *ce = gdn_option("GDN_PR_PLACE_COPIES");
  if (ce && ce[0] == '1') {
    phase = 1;
    if (rc == GDN_OK) rc = pl.rebase();
    for (int k = 0; k < p->n_mid_tiers && rc == GDN_OK; k++) rc = pl.search(p->mid[k].rec, "mid records");
    if (rc == GDN_OK && p->has_hub) rc = pl.search(p->hub_rec, "hub records");
    if (rc == GDN_OK) rc = pl.search(p->pb.V, "V");
    phase = 0;
    if (rc == GDN_OK) rc = pl.rebase();
    if (rc == GDN_OK) rc = pl.search(p->pb.U, "U");
  }
  pl.end();
  p->placing = false;
  if (rc != GDN_OK) return rc;
  unsigned zero = 0;  // the scratch iterations must not leave a range flag behind
  GDN_HIP(hipMemcpy(p->pb.errflag.p, &zero, sizeof(zero), hipMemcpyHostToDevice));
  return rc;
}

int gdn_pr_plan_check(gdn_pr_plan *plan) {
  GDN_REQUIRE(plan != nullptr, "plan");
  if (plan->tickets.p) {
    unsigned late = 0;
    GDN_HIP(hipMemcpy(&late, plan->tickets.p + (size_t)PB_TICKET_STRIDE * PB_MAX_PARTS, sizeof(late), hipMemcpyDeviceToHost));
    if (late) {
      gdn_set_error("gdn_pr_wait_part_dev: a part's tickets did not arrive within 4 s (was the pull it waits for ever queued?)");
      return GDN_ERR_HIP;
    }
  }
  if (plan->layout != GDN_LAYOUT_PB) return GDN_OK;
  unsigned f = 0;
  GDN_HIP(hipMemcpy(&f, plan->pb.errflag.p, sizeof(f), hipMemcpyDeviceToHost));
  if (f) {
    gdn_set_error("PB layout: a contribution outside [0,1] (or a row sum >= 2) reached the fixed-point accumulator; "
                  "scores must be a probability vector -- use GDN_LAYOUT_CSR for other inputs");
    return GDN_ERR_OVERFLOW;
  }
  return GDN_OK;
}

int gdn_pr_plan_refsum_info(const gdn_pr_plan *plan, int32_t *rows, int32_t *longest_row, uint64_t *entries, int32_t *groups) {
  GDN_REQUIRE(plan != nullptr, "plan");
  if (rows) *rows = (int32_t)plan->ref_n;
  if (longest_row) *longest_row = (int32_t)plan->ref_longest;
  if (entries) *entries = plan->ref_edges;
  if (groups) *groups = (!plan->ref_sum || plan->ref_n == 0) ? 0 : 4 + (plan->ref_n_vlong ? 1 : 0) + (plan->ref_n > plan->ref_n_vlong ? 1 : 0);
  return GDN_OK;
}

int gdn_pr_plan_layout(const gdn_pr_plan *plan, int32_t *layout, int32_t *log_blk) {
  GDN_REQUIRE(plan != nullptr, "plan");
  if (layout) *layout = plan->layout;
  if (log_blk) *log_blk = plan->layout == GDN_LAYOUT_PB ? plan->pb.log_chunk * 100 + plan->pb.log_bin : 0;
  return GDN_OK;
}

int gdn_pr_plan_hubs(const gdn_pr_plan *plan, int32_t *n_hubs, uint64_t *hub_edges) {
  GDN_REQUIRE(plan != nullptr, "plan");
  if (n_hubs) *n_hubs = plan->has_hub ? (int32_t)plan->n_hubs : 0;
  if (hub_edges) *hub_edges = plan->has_hub ? plan->hub.nnz : 0;
  return GDN_OK;
}

int gdn_pr_plan_bins(const gdn_pr_plan *plan, int32_t *n_bins) {
  GDN_REQUIRE(plan != nullptr && n_bins != nullptr, "null argument");
  *n_bins = plan->layout == GDN_LAYOUT_PB ? (int32_t)plan->pb.nbins : 0;
  return GDN_OK;
}

int gdn_pr_plan_mid(const gdn_pr_plan *plan, int32_t *n_tiers, int32_t *n_sources, uint64_t *n_edges) {
  GDN_REQUIRE(plan != nullptr, "plan");
  int32_t ns = 0;
  uint64_t ne = 0;
  for (int t = 0; t < plan->n_mid_tiers; t++) {
    ns += (int32_t)plan->mid[t].n;
    ne += plan->mid[t].layout.nnz;
  }
  if (n_tiers) *n_tiers = plan->n_mid_tiers;
  if (n_sources) *n_sources = ns;
  if (n_edges) *n_edges = ne;
  return GDN_OK;
}

uint64_t gdn_pr_iter_bytes(const gdn_pr_plan *plan) {
  if (!plan) return 0;
  // SURVEY 8d's definition on the graph the caller handed over (a squished plan still solves all m_orig vertices)
  const uint64_t m = (uint64_t)(plan->squished ? plan->m_orig : plan->m_local), nnz = plan->nnz;
  return 8 * (m + 1) + 4 * nnz + 4 * nnz + 16 * m;
}

// Host API: one call == PRSolver(g, scores) (src/pr/main.cc:19).
int gdn_pr(int32_t m, uint64_t nnz, const uint64_t *in_rowptr, const int32_t *in_colidx,
           const int32_t *out_degree, float *scores, float damping, double epsilon, int32_t max_iter,
           gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && in_rowptr && out_degree && scores, "null argument");
  GDN_REQUIRE(max_iter >= 1, "max_iter");
  GDN_TRY(gdn_require_device());
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  HostTimer th2d, tprep, tsolve;
  th2d.start();
  gdn_graph *g = nullptr;
  GDN_TRY(gdn_graph_upload(m, nnz, in_rowptr, in_colidx, &g));
  DevBuf<int32_t> d_deg;
  DevBuf<float> d_scores, d_state, d_c0, d_c1;
  DevBuf<double> d_diff;
  int rc = GDN_OK;
  gdn_pr_plan *plan = nullptr;
  do {
    if ((rc = d_deg.alloc(m)) || (rc = d_scores.alloc(m)) || (rc = d_diff.alloc(1))) break;
    if (hipMemcpy(d_deg.p, out_degree, (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_scores.p, scores, (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess) {
      gdn_set_error("gdn_pr: upload failed");
      rc = GDN_ERR_HIP;
      break;
    }
    st.h2d_ms = th2d.stop_ms();
    // graphs of the CSR regime: the whole solve in one cooperative launch (GDN_PR_FUSED=0: the per-iteration loop)
    {
      const char *env = gdn_option("GDN_PR_LAYOUT"), *fz = gdn_option("GDN_PR_FUSED");
      const char *rs = gdn_option("GDN_PR_SUM");  // (reference-order sums are a fix-up behind the per-iteration pull)
      const bool want = (rs && rs[0] == 'r') ? false : fz ? fz[0] != '0' : (!(env && (env[0] == 'c' || env[0] == 'p')) && nnz < (1ull << 18));  // measured: 20 us per iteration against the loop's 27 at 0.23 M edges, 35 against 30 at 1 M
      int fused = 0;
      if (want && (rc = pr_solve_fused(g, d_deg.p, d_scores.p, damping, epsilon, max_iter, &st, &fused))) break;
      if (fused) {
        g_pr_last_layout = GDN_LAYOUT_CSR;
        if (hipMemcpy(scores, d_scores.p, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) {
          gdn_set_error("gdn_pr: download failed");
          rc = GDN_ERR_HIP;
        }
        break;
      }
    }
    tprep.start();
    double pol_iters = 0, pol_csr_ms = 0, pol_pb_ms = 0;
    // the PB layout works on the live vertices only (GDN_LAYOUT_PB_SQUISHED; GDN_PR_SQUISH=0: the caller's vertex space)
    int32_t layout = GDN_LAYOUT_AUTO;
    {
      const char *env = gdn_option("GDN_PR_LAYOUT"), *sq = gdn_option("GDN_PR_SQUISH");
      // One call pays the layout it builds (the reference's blocked solvers do theirs before t.Start(),
      // src/pr/push_pb.cu:271,339 -- it is in prep_ms here, not in solve_ms, but it is wall time all the same).  Without a
      // forced layout the call picks by PREDICTED WALL TIME: iterations to epsilon ~ 24 at 1e-4 on power-law graphs (15 on
      // test/graphs/pr.mtx, 22 on RMAT-22), scaled with log(epsilon), capped by max_iter; per edge and iteration the
      // merge-path layout costs ~10 ps (18 ps from 2^28 edges on: the contribution vector no longer fits the Infinity
      // Cache), the blocked one 2.3 ps, and the blocked layout ~100 ps per edge to build with the tiered builder of round 4
      // (RMAT-22: 6.3 ms of prep for 65 M edges, RMAT-27: 0.15 s for 2.1 G; round 3: 480 ps -- the merge-path layout then won
      // every single solve at epsilon 1e-4) -- so the blocked layout wins from ~13 iterations on.  GDN_PR_ONESHOT=solve:
      // "blocked from 2^22 edges on" whatever the iteration count (best solve_ms, the number the reference's Timer prints);
      // GDN_PR_LAYOUT=c / p force a layout, anything else is this choice.
      const char *os_ = gdn_xoption("GDN_PR_ONESHOT");
      bool pb = nnz >= (1ull << 22);
      {
        double iters = 24.0;
        if (epsilon > 0.0 && epsilon < 1.0) iters = 24.0 * log(epsilon) / log(1e-4);
        if (!(epsilon > 0.0) || iters > (double)max_iter) iters = (double)max_iter;
        const double csr_ps = nnz >= (1ull << 28) ? 18.0 : 10.0, pb_ps = 2.3, build_ps = 100.0;
        if (pb && !(os_ && os_[0] == 's')) pb = iters * (csr_ps - pb_ps) > build_ps;
        // GDN_TRACE_POLICY=1: what the model predicted, next to what the solve then measured (the constants
        // were fitted to R-MAT on one box -- this line is how they are checked on another graph family)
        pol_iters = iters;
        pol_csr_ms = iters * csr_ps * (double)nnz * 1e-9;
        pol_pb_ms = (iters * pb_ps + build_ps) * (double)nnz * 1e-9;
      }
      const bool forced = env && (env[0] == 'c' || env[0] == 'p');
      if (pb && !forced && pr_gather_is_local(g, 0)) pb = false;  // lattice-like graphs: the merge-path layout is the faster one anyway
      if (forced) pb = env[0] == 'p';
      if (pb && !(sq && sq[0] == '0')) layout = GDN_LAYOUT_PB_SQUISHED;
      else if (!pb) layout = GDN_LAYOUT_CSR;
    }
    g_pr_no_place = true;
    rc = gdn_pr_plan_create(g, d_deg.p, m, 0, layout, &plan);
    g_pr_no_place = false;
    if (rc) break;
    g_pr_last_layout = plan->layout;
    int32_t ms = m;
    if ((rc = gdn_pr_plan_state_size(plan, &ms))) break;
    if ((rc = d_state.alloc(ms)) || (rc = d_c0.alloc(ms)) || (rc = d_c1.alloc(ms))) break;
    st.prep_ms = tprep.stop_ms();
    // timed region == src/pr/base.cu:110-128 (t.Start .. t.Stop around the do/while)
    tsolve.start();
    if ((rc = gdn_pr_import_dev(plan, d_scores.p, d_state.p, damping, nullptr))) break;
    double dead_diff = 0;
    if ((rc = gdn_pr_import_diff(plan, &dead_diff))) break;
    if ((rc = gdn_pr_contrib_dev(plan, d_state.p, d_c0.p, nullptr))) break;
    float *cin = d_c0.p, *cout = d_c1.p;
    int iter = 0;
    double diff = 0;
    g_pr_trace.clear();
    // Iterations are queued in BATCHES: the convergence test runs on the device behind every iteration
    // (pr_check_kernel) and, once it has fired, the pull launches queued behind it return at once (GdnSkippable), so the
    // state is the one of the converging iteration whatever the batch size -- and the host reads one word per batch
    // instead of blocking on the L1 change after every iteration (~20 us each: a sixth of an RMAT-22 iteration).
    int batch = 8;
    if (const char *e = gdn_option("GDN_PR_BATCH")) batch = atoi(e) > 0 ? atoi(e) : 1;  // tuning / test knob (1 = per iteration)
    DevBuf<PrLoopCtl> d_ctl;
    DevBuf<double> d_trace;
    if ((rc = d_ctl.alloc(1)) || (rc = d_trace.alloc((size_t)max_iter))) break;
    if (hipMemsetAsync(d_ctl.p, 0, sizeof(PrLoopCtl), 0) != hipSuccess) {
      gdn_set_error("gdn_pr: memset failed");
      rc = GDN_ERR_HIP;
      break;
    }
    plan->skip_flag = &d_ctl.p->done;
    PrLoopCtl h{0u, 0};
    int queued = 0;
    while (!h.done && queued < max_iter) {
      const int nb = std::min(batch, max_iter - queued);
      for (int k = 0; k < nb && !rc; k++) {
        rc = gdn_pr_pull_dev(plan, cin, d_state.p, cout, d_diff.p, damping, nullptr);
        hipLaunchKernelGGL(pr_check_kernel, dim3(1), dim3(64), 0, 0, d_diff.p, epsilon, dead_diff, d_ctl.p, d_trace.p);
        float *tmp = cin;
        cin = cout;
        cout = tmp;
      }
      queued += nb;
      if (rc) break;
      if (hipMemcpy(&h, d_ctl.p, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) {
        gdn_set_error("gdn_pr: loop state readback failed: %s", hipGetErrorString(hipGetLastError()));
        rc = GDN_ERR_HIP;
        break;
      }
    }
    plan->skip_flag = nullptr;
    if (rc) break;
    g_pr_trace.resize((size_t)h.n_iter);
    if (h.n_iter > 0 && hipMemcpy(g_pr_trace.data(), d_trace.p, (size_t)h.n_iter * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_pr: trace readback failed");
      rc = GDN_ERR_HIP;
      break;
    }
    diff = h.n_iter > 0 ? g_pr_trace.back() : 0.0;
    iter = h.done ? h.n_iter - 1 : max_iter;  // the reference's loop variable at exit (omp_base.cc:23-37)
    if (rc) break;
    if ((rc = gdn_pr_export_dev(plan, d_state.p, d_scores.p, damping, nullptr))) break;
    if (hipDeviceSynchronize() != hipSuccess) {
      gdn_set_error("gdn_pr: export failed: %s", hipGetErrorString(hipGetLastError()));
      rc = GDN_ERR_HIP;
      break;
    }
    st.solve_ms = tsolve.stop_ms();
    if ((rc = gdn_pr_plan_check(plan))) break;
    st.iterations = iter + 1;  // the reference prints iter+1 (omp_base.cc:39)
    st.last_error = diff;
    if (gdn_option("GDN_TRACE_POLICY"))
      fprintf(stderr, "[policy] gdn_pr: %llu edges, predicted %.0f iterations, merge-path %.2f ms, blocked %.2f ms (build included) -> %s;"
              " measured: %d iterations, prep %.2f + solve %.2f = %.2f ms (%.1f ps per edge and iteration, build %.0f ps per edge)\n",
              (unsigned long long)nnz, pol_iters, pol_csr_ms, pol_pb_ms, plan->layout == GDN_LAYOUT_CSR ? "merge-path" : "blocked",
              iter + 1, st.prep_ms, st.solve_ms, st.prep_ms + st.solve_ms,
              st.solve_ms * 1e9 / ((double)(nnz ? nnz : 1) * (iter + 1)), st.prep_ms * 1e9 / (double)(nnz ? nnz : 1));
    st.edges_traversed = nnz * (uint64_t)(iter < max_iter ? iter + 1 : max_iter);
    if (hipMemcpy(scores, d_scores.p, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_pr: download failed");
      rc = GDN_ERR_HIP;
    }
  } while (0);
  gdn_pr_plan_free(plan);
  gdn_graph_free(g);
  if (stats) *stats = st;
  return rc;
}

}  // extern "C"
