// gdn_pb.hpp -- propagation-blocked row reduction with LDS-resident slices (PageRank pull, SpMV; layout also
// used by the dense BFS / SSSP sweeps).
//
// Why: on MI355X a divergent 4-byte gather costs a vector-memory issue slot per cache line;
// tools/gather_probe measures 54 G gathers/s out of a 512 MiB table and ~200 G/s even when the
// table sits in L2 -- against 1.8 T dwords/s for coalesced streams.  A CSR pull iteration on
// RMAT-27 is 2^31 such gathers (38 ms, profiles/r01_csr_*).  tools/lds_probe: random
// ds_read_b32 runs at 4.2 T/s and ds_add_u64 at 2.9 T/s chip-wide (ds_add_f32 only 0.2 T/s).
// The reference's own answer to the gather wall is cache / propagation blocking
// (include/segmenting.h, include/prop_blocking.h, src/pr/push_pb.cu, src/pr/partition.cu);
// this is the CDNA4 form of it, built around the 160 KB LDS instead of a cache:
//
//   edges are grouped into TILES (source chunk c, destination bin b): chunk = up to 2^log_chunk
//   source ids (fp32 slice = 128 KB of LDS), bin = up to 2^log_bin destination rows (u64
//   accumulators = 128 KB of LDS); with `compact` the slices are cut over the vertices that occur
//   at all and spread over whole rounds of 256 workgroups.  Every tile is padded to `pad` edges
//   (PageRank / SpMV: 32 = whole 128-byte lines of vals).  Static data per (padded) edge:
//     U[e]  u16  source id within its chunk, CHUNK-major order (pad = chunk size -> a 0.0 slot)
//     V[e'] u16  destination row within its bin, BIN-major order (optionally 8-bit deltas: v8)
//     G[e >> log_group] u32 per group of edges in chunk-major order: its group index in bin-major
//   Per iteration:
//   phase A (pb_expand_kernel, one workgroup per chunk): x[chunk] -> LDS (coalesced, squeezed
//       through the activity bitmap); one flat sweep over the chunk's edges: 8-byte U load per lane,
//       4 LDS gathers, one 16-byte store into the bin-major vals array at G[group].
//   phase B (pb_accumulate_kernel, one workgroup per bin): ONE contiguous range of vals/V per
//       bin streamed with 16-/8-byte loads; each value is converted to 2^-62 fixed point and
//       added with ds_add_u64 (integer LDS atomics are 14x faster than float ones here, and
//       integer addition is associative, so the sums are BITWISE REPRODUCIBLE); then the fused
//       epilogue of the rows (coalesced, 16-byte accesses).  The edges of high-degree sources (RECORD TIERS: hubs and
//       up to two mid tiers, PbMidArgs) never pass through phase A: phase B reads them as 32-bit (source index, row)
//       records sorted by source and takes the value from the tier's per-iteration table (L2 resident).
//   PageRank sends fixed-point CODES through vals and the tables (pb_encode once per source, pb_decode per edge).
//   HBM traffic per main-layout edge ~ (2 + 0.125 + 4) + (4 + 2) = 12.1 B, per tier edge 4 B, all
//   streamed; no divergent vector-memory gather or scatter is left on the per-edge path.
//
// Fixed point: PageRank contributions lie in [0,1] and every row sum is <= the total rank mass
// <= 1 (no dangling redistribution in the reference, src/pr/omp_base.cc), so sum*2^62 fits a
// u64; a contribution >= 2^-38 is represented exactly (24-bit mantissa), smaller ones are
// truncated at 2^-62 = 2.2e-19 absolute.  That is MORE accurate than the reference's
// sequential fp32 accumulation.  Out-of-range inputs raise an error flag instead of wrapping.
#pragma once
#include <functional>
#include <vector>

#include "gdn_common.hpp"

#ifndef PB_THREADS
#define PB_THREADS 1024
#endif
#define PB_WAVES (PB_THREADS / 64)
#define PB_MAX_LOG_CHUNK 15  // 32768 floats = 128 KB of LDS
#define PB_MAX_LOG_BIN 14    // 16384 u64    = 128 KB of LDS
#define PB_FIX_SHIFT 62
#define PB_HR_THREADS 128    // threads of phase A that fold the hub-row edges while the others sweep
#ifndef PB_IL_QUADS
#define PB_IL_QUADS 2        // phase B, interleaved record streams: 16-byte record loads (= 4 table reads each) in flight per lane
#endif

// TICKETS (gdn_pr_pull_parts_dev): an accumulate launch whose bins run part by part (launch index < end[j] = part j; the
// order array lists the bins of part 0 first, largest first inside a part).  A workgroup that has finished its bin makes
// its rows visible device-wide and adds 1 to ticket[32 * j]; a one-wave kernel on ANOTHER stream waits for a part's
// count (pb_ticket_wait_kernel) -- the exchange of part j starts while parts j+1.. are still being accumulated, without
// cutting the iteration into launches whose tails leave the chip idle (4 launches: phase B 2.53 -> 3.13 ms on RMAT-27,
// profiles/r06_dist_one_rank.md).
#define PB_MAX_PARTS 8
#define PB_TICKET_STRIDE 32  // words: every counter on a 128-byte line of its own
struct PbParts {
  unsigned n = 0;
  unsigned end[PB_MAX_PARTS] = {};
  // how a workgroup's rows reach the other XCDs before its ticket: 0 = plain stores + one agent-scope release (an L2
  // write-back) per workgroup, 1 = the op stored them write-through (sc1; PrOp::wt) and only the waves' stores are awaited
  unsigned mode = 0;
};

struct PbPlan {
  bool v_il = false;     // V stored in lane-interleaved blocks of 512 edges (pb_v_interleave_kernel; every bin starts on a multiple of 512)
  int32_t m_local = 0;   // destination rows
  int32_t m_global = 0;  // source ids
  uint64_t nnz = 0;
  int log_chunk = 0, log_bin = 0;
  unsigned chunk_slots = 0;       // source slots used per chunk (<= 2^log_chunk) = the id of the zero slot pad edges read
  int log_group = 3;              // edges per entry of G = 2^log_group (tiles are padded to a multiple of it)
  uint32_t nchunks = 0, nbins = 0;
  uint64_t n_pad = 0;             // padded edge count (multiple of the group size) = length of U, V, vals
  DevBuf<uint16_t> U;             // chunk-major
  DevBuf<uint32_t> G;             // n_pad >> log_group (+ 1 dump group)
  DevBuf<uint16_t> V;             // bin-major (released when v8)
  // v8: the row of an edge = Vb[its 32-edge group] + the sum of the Vd bytes of the group up to and including it
  // (edges of a tile are sorted by row; filler edges with value 0 keep every distance <= 255): 1.06 B/edge, not 2
  bool v8 = false;
  DevBuf<uint8_t> Vd;             // bin-major, one byte per padded edge
  DevBuf<uint16_t> Vb;            // one base row per 32 padded edges
  DevBuf<float> vals;             // bin-major
  DevBuf<eoff_t> chunk_ptr;       // nchunks + 1, element units (multiples of 8)
  DevBuf<eoff_t> bin_ptr;         // nbins + 1, element units (multiples of 8)
  DevBuf<uint32_t> chunk_order;   // chunks by descending edge count (largest first: no long tail)
  DevBuf<uint32_t> bin_order;     // bins by descending edge count
  DevBuf<double> partial;         // nbins
  DevBuf<double> red_scratch;
  DevBuf<unsigned> errflag;       // 1 word: fixed-point range violation
  // vertex compaction (PageRank): chunks / bins are cut over the ACTIVE sources (referenced by an
  // edge) and ACTIVE rows (in-degree > 0) only -- on RMAT-27 half of the vertices are neither, so
  // tiles get 4x longer.  Bitmaps over the original ids + the original id range of every slice.
  bool compact = false;
  DevBuf<uint32_t> src_bits, dst_bits;   // ceil(m_global/32), ceil(m_local/32) words
  DevBuf<uint32_t> chunk_lo, bin_lo;     // nchunks + 1 / nbins + 1 original ids
  bool timing = false;
  std::vector<uint32_t> h_bin_lo;  // host copy of bin_lo (compact layouts): first original row of every bin
  std::vector<hipEvent_t> ev;  // triples: start, after A, after B
  size_t ev_used = 0;
  ~PbPlan() {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }
};

// The two 8-byte-per-edge key buffers of a layout build, kept across the builds of ONE plan: a fresh 17 GB hipMalloc
// costs ~1 s at RMAT-27 (page mapping), and a PageRank plan builds four layouts.
struct PbScratch {
  DevBuf<unsigned long long> ka, kb;
};

// PLACEMENT SEARCH of a plan's streamed arrays.  Where hipMalloc puts them moves a sweep by up to 9 % (PageRank, RMAT-27:
// 3.77 .. 4.15 ms between plans of ONE process that differ in nothing else; phase A follows `vals` (1.04 .. 1.23 ms) and a
// little `U`, phase B follows `V` and the record streams, 0.03 .. 0.1 ms each -- tools/pr_place_probe.py,
// profiles/r03_pb_placement.txt).  Physical addresses are not visible to an unprivileged process, and neither staggered bases
// nor shuffled 2 MiB chunks changed the spread.  So a plan measures: array by array, up to `tries` fresh
// allocations are timed on scratch vectors (`timed`: the mean of a few sweeps), the fastest one stays; a rejected allocation is
// held until the array is done, else hipMalloc would hand the same block out again.  Values do not matter to the timing, and
// no result depends on where an array lives.  Bounded by budget_ms (a fresh allocation costs ~50 ms per GB of page mapping
// once the blocks the plan build freed are used up).  What does NOT explore the spread: offsets inside one allocation (16
// random 4 KiB multiples inside a 64 MiB slack: all within 0.5 % of each other, the other arena 2 % away) -- it is the
// physical region, not the alignment.
struct PbPlacer {
  int tries = 0;
  double budget_ms = 0;
  const char *tag = "";
  bool trace = false;
  size_t min_bytes = (size_t)64 << 20;  // smaller arrays live in the caches (GDN_PLACE_MIN_BYTES: tests force the search)
  std::function<int(double *)> timed;
  HostTimer wall;
  double best = 0, first = 0;
  int begin() {
    if (const char *e = gdn_test_option("GDN_PLACE_MIN_BYTES")) min_bytes = (size_t)strtoull(e, nullptr, 10);
    GDN_HIP(hipDeviceSynchronize());
    wall.start();
    GDN_TRY(timed(&best));  // (the first one also loads code objects in a fresh process)
    GDN_TRY(timed(&best));
    first = best;
    return GDN_OK;
  }
  template <typename T>
  int search(DevBuf<T> &buf, const char *name, int mult = 1) {
    if (!buf.p || buf.n * sizeof(T) < min_bytes) return GDN_OK;
    std::vector<DevBuf<T> *> held;
    int rc = GDN_OK;
    for (int k = 0; k < tries * mult && rc == GDN_OK; k++) {
      if (wall.stop_ms() > budget_ms) break;
      // at most two rejected allocations are held at a time (six extra copies of a 4.4 GB array otherwise)
      if (held.size() >= 2) {
        delete held.front();
        held.erase(held.begin());
      }
      DevBuf<T> *old = new DevBuf<T>();
      held.push_back(old);
      if (buf.move(old) != GDN_OK) {  // no memory for a second copy: the array stays where it is, and that is no error
        gdn_set_error("%s", "");
        break;
      }
      double cur = 0;
      if ((rc = timed(&cur)) != GDN_OK) break;
      if (trace) fprintf(stderr, "[%s place] %-12s try %d: %.3f ms (best %.3f) at %p (was %p)\n", tag, name, k, cur, best, (void *)buf.p, (void *)old->p);
      if (cur < best * 0.997) best = cur;
      else buf.swap(*old);  // back to the faster allocation; the new one is held until the array is done, then freed
    }
    for (DevBuf<T> *h : held) delete h;
    return rc;
  }
  // a new baseline (the caller changed what `timed` measures: another phase)
  int rebase() { return timed(&best); }
  // the same for an array whose CONTENTS are scratch of one iteration (PageRank's vals: written by phase A, read by phase B):
  // a candidate is a fresh allocation, nothing is copied.  Rejected candidates stay allocated until the array is done (hipMalloc
  // hands a freed block out again), at most `hold` of them.
  template <typename T>
  int search_fresh(DevBuf<T> &buf, const char *name, int n_tries, int hold, double stop_ratio = 0.0) {
    if (!buf.p || buf.n * sizeof(T) < min_bytes) return GDN_OK;
    std::vector<DevBuf<T> *> held;
    int rc = GDN_OK;
    double worst = best;  // slowest placement seen: a candidate below stop_ratio x that IS one of the fast blocks -- stop there
    for (int k = 0; k < n_tries && rc == GDN_OK; k++) {
      if (wall.stop_ms() > budget_ms) break;
      if (stop_ratio > 0.0 && best < stop_ratio * worst) break;
      if ((int)held.size() >= hold) {
        delete held.front();
        held.erase(held.begin());
      }
      DevBuf<T> *cand = new DevBuf<T>();
      held.push_back(cand);
      // GDN_PLACE_OFFSETS=1 (diagnostic, profiles/r05_pb_place_offsets.txt): every candidate is 256 MB longer than the array and is
      // timed at several offsets inside itself as well -- does a placement's speed belong to the ALLOCATION (its pages) or to
      // the ADDRESS (an interleaving of its bits)?
      const bool probe_offsets = gdn_xoption("GDN_PLACE_OFFSETS") != nullptr;
      const size_t slack = probe_offsets ? ((size_t)256 << 20) / sizeof(T) : 0;
      // (optional: never at the price of the scratch cache or of more than half of the free memory; when memory is
      // short the oldest held candidates go first, and a search that still finds none ends without an error)
      bool got = cand->alloc_optional(buf.n + slack);
      while (!got && held.size() > 1) {
        delete held.front();
        held.erase(held.begin());
        got = cand->alloc_optional(buf.n + slack);
      }
      if (!got) break;
      cand->n = buf.n;
      // (slots no launch writes -- alignment gaps between bins -- must read as zero, as in the array the builder made)
      if (hipMemsetAsync(cand->p, 0, (buf.n + slack) * sizeof(T), 0) != hipSuccess) {
        (void)hipGetLastError();
        break;
      }
      buf.swap(*cand);
      double cur = 0;
      if ((rc = timed(&cur)) != GDN_OK) break;
      if (probe_offsets) {
        T *const base = buf.p;
        const size_t offs[] = {(size_t)4 << 10, (size_t)64 << 10, (size_t)2 << 20, ((size_t)2 << 20) + ((size_t)64 << 10), (size_t)32 << 20,
                               (size_t)128 << 20, (size_t)256 << 20};
        for (size_t ob : offs) {
          buf.p = base + ob / sizeof(T);
          double t = 0;
          if ((rc = timed(&t)) != GDN_OK) break;
          fprintf(stderr, "[%s place] %-12s fresh %d at %p + %9zu B: %.3f ms (at + 0: %.3f)\n", tag, name, k, (void *)base, ob, t, cur);
        }
        buf.p = base;
        if (rc != GDN_OK) break;
      }
      if (trace) fprintf(stderr, "[%s place] %-12s fresh %d: %.3f ms (best %.3f) at %p (was %p)\n", tag, name, k, cur, best, (void *)buf.p, (void *)cand->p);
      if (cur > worst) worst = cur;
      if (cur < best * 0.997) best = cur;
      else buf.swap(*cand);
    }
    if (gdn_xoption("GDN_PLACE_OFFSETS") && rc == GDN_OK) {
      // ... and is it a property of the allocation at all, or of the MOMENT it was timed in?  Every candidate still held is
      // timed once more, now that the search is over (seconds after the first ones were timed)
      for (size_t i = 0; i < held.size() && rc == GDN_OK; i++) {
        if (!held[i]->p || held[i]->p == buf.p) continue;
        buf.swap(*held[i]);
        double t = 0;
        rc = timed(&t);
        fprintf(stderr, "[%s place] %-12s held candidate at %p timed again at the end: %.3f ms\n", tag, name, (void *)buf.p, t);
        buf.swap(*held[i]);
      }
    }
    for (DevBuf<T> *h : held) delete h;
    return rc;
  }
  void end() {
    (void)hipDeviceSynchronize();
    if (trace) fprintf(stderr, "[%s place] %.3f -> %.3f ms per sweep, %.0f ms spent\n", tag, first, best, wall.stop_ms());
  }
};

// implemented in gdn_build.hip (uses the radix sort)
// alloc_vals = false: only the static layout (U, V, G, pointers, orders) -- BFS keeps 1 bit per edge
// edge_vals_in (nullable, CSR order) -> *edge_vals_out in chunk-major order, pads = 0 (SpMV's Ax)
int pb_build(const gdn_graph *in_csr, int32_t m_global, int log_chunk, int log_bin, PbPlan &p, bool alloc_vals = true,
             const float *edge_vals_in = nullptr, DevBuf<float> *edge_vals_out = nullptr, bool compact = false,
             bool rows_are_sources = false,  // true: `in_csr` is an OUT-CSR (row = source, col = destination)
             unsigned pad = 16,              // every tile is padded to a multiple of `pad` edges (power of two <= 128)
             int log_group = 3,              // 2^log_group <= pad edges share one entry of G
             const uint8_t *src_class = nullptr,  // per global source id: only edges whose source has class
             int want_class = 0,                  //   `want_class` are laid out (PageRank hub tier, gdn_pr.hip)
             bool src_major = false,              // order inside a tile: (source, row) instead of (row, source)
             bool v_delta = false,                // rows as 8-bit distances (Vd, Vb) instead of u16 (V): see PbPlan::v8
             const uint8_t *dst_class = nullptr,  // per row: only edges whose row has class `want_dst` are laid out
             int want_dst = 0,
             bool rows_of_class_only = false,     // the row slices hold the rows of that class only (hub-row layout)
             bool no_gaps = false,                // slice starts padded to `pad` only: with ONE bin (or chunk) U and V
                                                  // then sit at the same positions
             int bin_balance_log = PB_MAX_LOG_BIN,  // bins of this size are spread over whole rounds of workgroups
             struct PbScratch *scratch = nullptr);   // key buffers shared by the builds of one plan (see PbScratch)

// Hub tier (gdn_pr.hip, gdn_spmv.hip): the edges of the <= 2^15 sources with the most out-edges live in a second layout
// (one source chunk, tiles sorted by hub) that phase B reads directly; their source values come from a
// PB_HUB_SLOTS-entry table refreshed per multiply (slot 2^15 = 0 for pad edges).
#define PB_HUB_LOG 15
#define PB_HUB_SLOTS ((1 << PB_HUB_LOG) + 1)
// picks the hub sources of `in_csr` (gdn_build.hip): cls gets one byte per source id, hub_ids the ascending ids;
// *n_hubs == 0: no hub tier
int pb_pick_hubs(const gdn_graph *in_csr, int32_t m_global, int log_bin, DevBuf<uint8_t> &cls, DevBuf<uint32_t> &hub_ids,
                 unsigned *n_hubs);

// Mid tiers (gdn_pr.hip): below the hubs, the sources that still own a fraction of an edge per average bin (R-MAT: the
// next two degree levels, a third of all edges).  Their edges do not pass through phase A / vals either: phase B reads
// them per bin as ONE stream of 32-bit records (source index in the tier << 14 | row in the bin), sorted by source, one
// record per lane, and fetches the value from the tier's table (<= 1 MB, refreshed per iteration, L2 resident).  Sorted
// by source and one record per lane, the 64 gathers of a wave instruction fall into a handful of consecutive cache
// lines: a near-coalesced load, not a divergent gather.  4 B/edge of HBM traffic instead of 12.1.
#ifndef PB_MAX_MID
#define PB_MAX_MID 4  // PageRank's default (SpMV keeps 2: its records carry Ax, 8 B per edge); profiles/r03_pb_tier_sweep*.txt
#endif
#define PB_MID_ROW_BITS 14
#define PB_MID_MAX ((1u << (32 - PB_MID_ROW_BITS)) - 1u)  // sources per tier; index PB_MID_MAX can be the zero slot
#define PB_MAX_REC_TIERS (1 + PB_MAX_MID)  // record streams of phase B: PageRank's hubs + the mid tiers
struct PbMidArgs {  // kernel argument of phase B
  const eoff_t *ptr[PB_MAX_REC_TIERS];     // nbins + 1 record offsets per tier (multiples of 16)
  const uint32_t *rec[PB_MAX_REC_TIERS];
  const float *val[PB_MAX_REC_TIERS];      // n + 1 values (slot n = 0 for pad records)
  unsigned zrec[PB_MAX_REC_TIERS];         // the pad record: n << PB_MID_ROW_BITS
  int form[PB_MAX_REC_TIERS];              // 0 record per lane, 1 four records per lane + table window (val: n + 4 slots),
                                           // 2 lane-interleaved blocks of 256 records: a 16-byte load = records l + 64 j
  const float *A[PB_MAX_REC_TIERS];        // nullable: per-record factor (SpMV's Ax in record order): value = table * A
  int n = 0;
  int v_il = 0;                            // PbPlan::v_il: V in lane-interleaved blocks of 512 edges (one 16-byte load = the rows of two quads)
};
struct PbTierRefresh {  // kernel argument of phase A (PageRank): the tier tables are refreshed by the same launch
  const uint32_t *ids[PB_MAX_REC_TIERS];  // original id of source k
  float *val[PB_MAX_REC_TIERS];           // slots[t] entries: the code of x[ids[k]] for k < n[t], 0 behind
  unsigned n[PB_MAX_REC_TIERS], slots[PB_MAX_REC_TIERS];
  int ntiers = 0;
  const unsigned *skip = nullptr;  // see GdnSkippable (gdn_common.hpp): a non-zero word = this launch does nothing
};
// min16: a mid-tier source owns at least min16 / 16 edges per average bin (0 = PB_MID_MIN_PER_BIN16; GDN_PB_MID_MIN16 overrides)
int pb_pick_tiers(const gdn_graph *in_csr, int32_t m_global, int log_bin, DevBuf<uint8_t> &cls, DevBuf<uint32_t> &hub_ids,
                  unsigned *n_hubs, int max_mid, DevBuf<uint32_t> *mid_ids, unsigned *n_mid, unsigned min16 = 0);
// turns the layout pb_build made for one mid class (log_chunk 15, src_major, same bins as the main layout) into the
// bin-major record stream; releases U, V and G of the layout
// ev (nullable): the layout's per-edge values (pb_build's edge_vals_out, chunk-major) are moved into record order
int pb_mid_finish(PbPlan &layout, unsigned n_src, DevBuf<uint32_t> &rec, DevBuf<float> *ev = nullptr);
// launch order of phase B: largest first by ALL the bytes of a bin -- main stream, the record streams of the tiers
// (tier_bin_ptr[t]: device array of nbins + 1 record offsets), the rows of its original range
int pb_order_bins_by_work(PbPlan &main, int n_tiers, const eoff_t *const *tier_bin_ptr, double main_bytes_per_edge,
                          double rec_bytes, double row_bytes);

// ---- round 4: the main layout AND its record tiers from one gather pass over the edges (gdn_pbtier.hpp, built into
// gdn_build.hip).  Replaces pb_pick_tiers + one pb_build per layout + pb_mid_finish for in-CSR plans.
struct PbTierSet {
  int n = 0;
  struct Tier {
    unsigned n_src = 0;
    uint64_t nnz = 0;
    DevBuf<uint32_t> ids;     // original (label) id of source k, ascending
    DevBuf<uint32_t> rec;     // bin-major records (source index << 14 | row), pad records = n_src << 14
    DevBuf<eoff_t> bin_ptr;   // nbins + 1 record offsets (multiples of 16; of 256 when interleaved)
    bool interleaved = false; // blocks of 256 records stored lane-interleaved (pt_interleave_kernel): phase B form 2
    DevBuf<float> A;          // PbTieredArgs::edge_vals: the records' values, in record order (pad records: 0)
  } t[PB_MAX_REC_TIERS];
  bool first_is_hub = false;  // t[0] is the hub tier (<= 2^15 sources), else the mid tiers start at t[0]
};

struct PbTieredArgs {
  const eoff_t *rowptr = nullptr;   // m_rows + 1, rows in the layout's vertex space
  const vid_t *colidx = nullptr;    // caller's column ids ...
  const eoff_t *colmap = nullptr;   // ... and (nullable) their map into the layout's vertex space: m_raw + 1 entries, the
  int32_t m_raw = 0;                //     exclusive scan of the live flags (needs src_count)
  int32_t m_rows = 0, m_global = 0;
  uint64_t nnz = 0;
  const int32_t *src_count = nullptr;  // nullable, m_global: out-edge count of every source (> 0 for every column that occurs)
  int log_chunk = 15, log_bin = 14;
  unsigned pad = 32;
  int log_group = 5;
  bool tiers = true;
  int max_mid = PB_MAX_MID;
  unsigned min16 = 1;
  int bin_balance_log = PB_MAX_LOG_BIN;
  bool alloc_vals = true;
  bool interleave = false;  // mid-tier record streams in lane-interleaved blocks of 256 (PbTierSet::Tier::interleaved)
  bool v_interleave = false;  // V in lane-interleaved blocks of 512 edges when every bin starts on a multiple of 512 (PbPlan::v_il)
  // one 32-bit value per edge in CSR order (SpMV's Ax, nullable) and where the main layout's go: (*main_vals)[k] belongs to
  // the edge whose source slot is U[k] (chunk-major tile order, 0 in the padding); the tiers' go to PbTierSet::Tier::A
  const float *edge_vals = nullptr;
  DevBuf<float> *main_vals = nullptr;
};

// GDN_OK; 1 = shape outside the builder's limits (nothing built: use pb_build); 2 = a column occurs whose src_count is 0
// (repeat with src_count = nullptr); < 0 error
int pb_build_tiered_run(const PbTieredArgs &a, PbPlan &p, PbTierSet &ts);

// ---- the same for an OUT-CSR (rows = sources) with one 32-bit value per edge: SSSP's blocked layout and its record tiers
struct PbOutTiers {  // gdn_sssp.hip's record tiers (see gdn_sssp_plan)
  int n = 0;
  unsigned off[PB_MAX_REC_TIERS + 1] = {};  // first source of tier t in ids
  DevBuf<uint32_t> ids;                     // tier sources, tier by tier, ascending ids inside a tier
  DevBuf<uint32_t> rec;                     // records (index in the tier << 15 | row), tier-major then bin-major
  DevBuf<uint8_t> w8;                       // their weights (nullable)
  DevBuf<eoff_t> ptr;                       // n x nbins + 1 (interleaved: every stream starts on a multiple of 256 records)
  DevBuf<uint32_t> cnt;                     // interleaved only: n x nbins record counts (the streams are not padded with records)
  bool interleaved = false;                 // the whole 256-record blocks of every stream (and of w8) are lane-interleaved
  unsigned long long edges = 0;
};
struct PbOutArgs {
  const gdn_graph *g = nullptr;   // out-CSR
  const int32_t *weight = nullptr;
  int log_chunk = 15, log_bin = 15;
  unsigned pad = 32;
  int log_group = 3;
  int max_tiers = 0;              // 0: no record tiers
  unsigned tier_min_deg = 8;
  unsigned caps[PB_MAX_REC_TIERS] = {1u << 15, 1u << 17, 1u << 17, 1u << 17, 1u << 17};
  bool want_w8 = false;
  bool interleave = false;        // record streams (and w8) in lane-interleaved blocks of 256, see pt_interleave_kernel
};

// GDN_OK; 1 = outside the builder's limits (nothing built: pb_build + sssp_build_tiers)
int pb_build_out_tiered_run(const PbOutArgs &a, PbPlan &p, DevBuf<float> &Wp, PbOutTiers &ts);

// the rows with the most in-edges (gdn_build.hip): at most max_rows rows with >= min_deg in-edges each
uint64_t pb_slots_per_slice(uint64_t n_act, int lg, int lg_full);  // vertices per slice after round balancing
int pb_pick_hub_rows(const gdn_graph *in_csr, unsigned max_rows, uint64_t min_deg, DevBuf<uint8_t> &dcls,
                     DevBuf<uint32_t> &row_ids, unsigned *n_rows);

#ifdef __HIPCC__
typedef unsigned short pb_u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short pb_u16x4 __attribute__((ext_vector_type(4)));
typedef float pb_f32x4 __attribute__((ext_vector_type(4)));
typedef int pb_i32x4 __attribute__((ext_vector_type(4)));

// plain-float table of a record tier (SpMV): val[k] = x[ids[k]] for k < n, 0 up to n_slots
static __global__ void __launch_bounds__(GDN_BLOCK)
pb_tier_gather_f32_kernel(const float *__restrict__ x, const uint32_t *__restrict__ ids, unsigned n, unsigned n_slots,
                          float *__restrict__ val, unsigned *__restrict__ absmax = nullptr) {
  // grid-stride (the launch uses few workgroups): with absmax, ONE atomic per workgroup
  __shared__ unsigned s_mx[GDN_WAVES_PER_BLOCK];
  unsigned mx = 0u;
  for (unsigned k = blockIdx.x * GDN_BLOCK + threadIdx.x; k < n_slots; k += gridDim.x * GDN_BLOCK) {
    const float v = k < n ? x[ids[k]] : 0.0f;
    val[k] = v;
    const unsigned bts = __float_as_uint(v) & 0x7FFFFFFFu;
    mx = (bts > mx && bts < 0x7F800000u) ? bts : mx;  // finite values only: inf / nan products are repaired row by row
  }
  if (absmax) {  // max |x| of the tier's columns (float bits), see pb_expand_scaled_kernel
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned t = (unsigned)__shfl_xor((int)mx, o, 64);
      mx = t > mx ? t : mx;
    }
    if (gdn_lane() == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned t = 0u;
      for (int w = 0; w < GDN_WAVES_PER_BLOCK; w++) t = s_mx[w] > t ? s_mx[w] : t;
      if (t) atomicMax(absmax, t);
    }
  }
}


// exclusive scan over the PB_THREADS threads of a workgroup; scratch = PB_WAVES + 1 unsigned
__device__ __forceinline__ unsigned pb_block_excl_scan(unsigned v, unsigned *scratch, unsigned *total) {
  const unsigned incl = gdn_wave_incl_scan(v);
  const unsigned w = threadIdx.x >> 6;
  __syncthreads();
  if (gdn_lane() == 63) scratch[w] = incl;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < PB_WAVES; i++) {
    const unsigned x = scratch[i];
    if ((unsigned)i < w) base += x;
    tot += x;
  }
  *total = tot;
  return base + incl - v;
}

// Walk the original ids [lo,hi) of a compacted slice in tiles of 32*PB_THREADS ids.  For every tile
// s_bits[w] / s_pref[w] (PB_THREADS words each) hold the activity word and the compact index of its
// first active id; fn(id, k) is called for every ACTIVE id (k = index inside the slice) and
// fn_inactive(id) for the others, ids visited in coalesced order.
template <class FA, class FI>
__device__ __forceinline__ unsigned pb_walk_slice(const uint32_t *__restrict__ bits, unsigned lo, unsigned hi,
                                              unsigned *s_bits, unsigned *s_pref, unsigned *s_scr, FA fn, FI fn_inactive) {
  unsigned running = 0;
  const unsigned w_begin = lo >> 5, w_end = (hi + 31u) >> 5;
  for (unsigned wb = w_begin; wb < w_end; wb += PB_THREADS) {
    const unsigned w = wb + threadIdx.x;
    unsigned b = (w < w_end) ? bits[w] : 0u;
    if (w == w_begin && (lo & 31u)) b &= ~0u << (lo & 31u);
    if (w + 1 == w_end && (hi & 31u)) b &= ~0u >> (32u - (hi & 31u));
    unsigned total;
    const unsigned ex = pb_block_excl_scan((unsigned)__popc(b), s_scr, &total);
    s_bits[threadIdx.x] = b;
    s_pref[threadIdx.x] = running + ex;
    __syncthreads();
    const unsigned id0 = wb << 5;
#pragma unroll 4
    for (unsigned j = 0; j < 32; j++) {
      const unsigned off = j * PB_THREADS + threadIdx.x;
      const unsigned id = id0 + off;
      if (id >= lo && id < hi) {
        const unsigned bw = s_bits[off >> 5];
        const unsigned bit = 1u << (off & 31u);
        if (bw & bit) fn(id, s_pref[off >> 5] + (unsigned)__popc(bw & (bit - 1u)));
        else fn_inactive(id);
      }
    }
    running += total;
    __syncthreads();
  }
  return running;  // active ids of the slice (the same in every thread)
}

// Epilogue walk over the original rows [lo,hi) of a bin with software prefetch: per step every thread
// gathers the row state of EPI rows (op.pre: independent loads in flight together), then finishes
// them (op.fin: compute + stores).  bits == nullptr: plain bin (row r <-> accumulator r - lo).
#define PB_EPI 8
template <class Op, class SumOf>
__device__ __forceinline__ double pb_epilogue(const uint32_t *__restrict__ bits, unsigned lo, unsigned hi,
                                              unsigned *s_bits, unsigned *s_pref, unsigned *s_scr, const Op &op,
                                              SumOf sum_of) {
  double dsum = 0.0;
  unsigned running = 0;
  const unsigned w_begin = lo >> 5, w_end = (hi + 31u) >> 5;
  for (unsigned wb = w_begin; wb < w_end; wb += PB_THREADS) {
    const unsigned w = wb + threadIdx.x;
    unsigned b = 0u;
    if (w < w_end) b = bits ? bits[w] : ~0u;
    if (w == w_begin && (lo & 31u)) b &= ~0u << (lo & 31u);
    if (w + 1 == w_end && (hi & 31u)) b &= ~0u >> (32u - (hi & 31u));
    unsigned total;
    const unsigned ex = pb_block_excl_scan((unsigned)__popc(b), s_scr, &total);
    s_bits[threadIdx.x] = b;
    s_pref[threadIdx.x] = running + ex;
    __syncthreads();
    const unsigned id0 = wb << 5;
    for (unsigned j0 = 0; j0 < 32; j0 += PB_EPI) {
      typename Op::Pre pre[PB_EPI];
      float sum[PB_EPI];
      bool ok[PB_EPI];
#pragma unroll
      for (int j = 0; j < PB_EPI; j++) {
        const unsigned off = (j0 + j) * PB_THREADS + threadIdx.x;
        const unsigned id = id0 + off;
        ok[j] = id >= lo && id < hi;
        sum[j] = 0.0f;
        if (ok[j]) {
          const unsigned bw = s_bits[off >> 5];
          const unsigned bit = 1u << (off & 31u);
          if (bw & bit) sum[j] = sum_of(s_pref[off >> 5] + (unsigned)__popc(bw & (bit - 1u)));
          pre[j] = op.pre((int32_t)id);
        }
      }
#pragma unroll
      for (int j = 0; j < PB_EPI; j++) {
        const unsigned id = id0 + (j0 + j) * PB_THREADS + threadIdx.x;
        if (ok[j]) dsum += op.fin((int32_t)id, sum[j], pre[j]);
      }
    }
    running += total;
    __syncthreads();
  }
  return dsum;
}

// 16-byte form of pb_walk_slice for phase A's prologue: s_x[k] = x[id] for the ACTIVE ids of [lo,hi).
// A tile is 32*PB_THREADS ids = 8*PB_THREADS groups of 4 ids; every thread issues the loads of its 8
// groups back to back (one HBM round trip per tile instead of eight), then squeezes them into LDS
// through the activity nibble.  x must be 16-byte aligned; ids beyond m_global are never active.
__device__ __forceinline__ unsigned pb_load_slice4(const float *__restrict__ x, int32_t m_global,
                                               const uint32_t *__restrict__ bits, unsigned lo, unsigned hi, float *s_x,
                                               unsigned *s_bits, unsigned *s_pref, unsigned *s_scr) {
  unsigned running = 0;
  const unsigned w_begin = lo >> 5, w_end = (hi + 31u) >> 5;
  const pb_f32x4 *x4 = reinterpret_cast<const pb_f32x4 *>(x);
  for (unsigned wb = w_begin; wb < w_end; wb += PB_THREADS) {
    const unsigned w = wb + threadIdx.x;
    unsigned b = (w < w_end) ? bits[w] : 0u;
    if (w == w_begin && (lo & 31u)) b &= ~0u << (lo & 31u);
    if (w + 1 == w_end && (hi & 31u)) b &= ~0u >> (32u - (hi & 31u));
    unsigned total;
    const unsigned ex = pb_block_excl_scan((unsigned)__popc(b), s_scr, &total);
    s_bits[threadIdx.x] = b;
    s_pref[threadIdx.x] = running + ex;
    __syncthreads();
    const unsigned g0 = wb << 3;  // first group of 4 ids of this tile
    pb_f32x4 v[8];
    unsigned nib[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const unsigned g = (unsigned)j * PB_THREADS + threadIdx.x;
      nib[j] = (s_bits[g >> 3] >> ((g & 7u) * 4u)) & 15u;
      if (nib[j]) {
        const size_t id = ((size_t)g0 + g) << 2;
        if (id + 3 < (size_t)m_global) v[j] = __builtin_nontemporal_load(x4 + g0 + g);
        else {
          v[j].x = x[id];  // an active id is < m_global; the rest of the group may not be
          v[j].y = id + 1 < (size_t)m_global ? x[id + 1] : 0.0f;
          v[j].z = id + 2 < (size_t)m_global ? x[id + 2] : 0.0f;
          v[j].w = 0.0f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (nib[j]) {
        const unsigned g = (unsigned)j * PB_THREADS + threadIdx.x;
        const unsigned sh = (g & 7u) * 4u;
        unsigned k = s_pref[g >> 3] + (unsigned)__popc(s_bits[g >> 3] & ((1u << sh) - 1u));
        if (nib[j] & 1u) s_x[k++] = v[j].x;
        if (nib[j] & 2u) s_x[k++] = v[j].y;
        if (nib[j] & 4u) s_x[k++] = v[j].z;
        if (nib[j] & 8u) s_x[k] = v[j].w;
      }
    }
    running += total;
    __syncthreads();
  }
  return running;
}

// 16-byte form of pb_epilogue (Op::vec_ok: every row array of the op is 16-byte aligned): a thread owns
// groups of 4 consecutive rows, prefetches the row state of PB_EPI4 groups with 16-byte loads
// (op.pre4), then finishes them with 16-byte stores (op.fin4).  Groups cut by lo / hi take the scalar
// path of op.pre / op.fin.
#define PB_EPI4 4
template <class Op, class SumOf>
__device__ __forceinline__ double pb_epilogue4(const uint32_t *__restrict__ bits, unsigned lo, unsigned hi,
                                               unsigned *s_bits, unsigned *s_pref, unsigned *s_scr, const Op &op,
                                               SumOf sum_of) {
  double dsum = 0.0;
  unsigned running = 0;
  const unsigned w_begin = lo >> 5, w_end = (hi + 31u) >> 5;
  for (unsigned wb = w_begin; wb < w_end; wb += PB_THREADS) {
    const unsigned w = wb + threadIdx.x;
    unsigned b = 0u;
    if (w < w_end) b = bits ? bits[w] : ~0u;
    if (w == w_begin && (lo & 31u)) b &= ~0u << (lo & 31u);
    if (w + 1 == w_end && (hi & 31u)) b &= ~0u >> (32u - (hi & 31u));
    unsigned total;
    const unsigned ex = pb_block_excl_scan((unsigned)__popc(b), s_scr, &total);
    s_bits[threadIdx.x] = b;
    s_pref[threadIdx.x] = running + ex;
    __syncthreads();
    const unsigned id0 = wb << 5;
    const unsigned tile_hi = (w_end - wb < PB_THREADS ? w_end - wb : PB_THREADS) << 5;  // ids of this tile that can be in range
    for (unsigned j0 = 0; j0 < 8; j0 += PB_EPI4) {
      if (((j0 * PB_THREADS) << 2) >= tile_hi) break;  // uniform: the rest of the tile is past hi
      typename Op::Pre4 pre[PB_EPI4];
      int kind[PB_EPI4];  // 0 nothing, 1 whole group, 2 cut group
#pragma unroll
      for (int j = 0; j < PB_EPI4; j++) {
        const unsigned id = id0 + ((((unsigned)(j0 + j)) * PB_THREADS + threadIdx.x) << 2);
        kind[j] = (id >= lo && id + 3 < hi && id + 3 > id) ? 1 : ((id + 3 >= lo && id < hi) ? 2 : 0);
        if (kind[j] == 1) pre[j] = op.pre4((int32_t)id);
      }
#pragma unroll
      for (int j = 0; j < PB_EPI4; j++) {
        if (!kind[j]) continue;
        const unsigned g = ((unsigned)(j0 + j)) * PB_THREADS + threadIdx.x;
        const unsigned id = id0 + (g << 2);
        const unsigned sh = (g & 7u) * 4u;
        const unsigned bw = s_bits[g >> 3];
        const unsigned nb = (bw >> sh) & 15u;
        unsigned k = s_pref[g >> 3] + (unsigned)__popc(bw & ((1u << sh) - 1u));
        float sum[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          sum[c] = 0.0f;
          if (nb & (1u << c)) sum[c] = sum_of(k++);
        }
        if (kind[j] == 1) dsum += op.fin4((int32_t)id, sum, pre[j]);
        else {
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const unsigned r = id + (unsigned)c;
            if (r >= lo && r < hi) dsum += op.fin((int32_t)r, sum[c], op.pre((int32_t)r));
          }
        }
      }
    }
    running += total;
    __syncthreads();
  }
  return dsum;
}

// float in [0,1] -> 2^-62 fixed point (truncating); flags anything else.  Branch-free: the 24-bit mantissa is
// placed at bits 39..62 (the image of 1.0) and shifted right by 127 - exponent, clamped to 63 (a 64-bit shift only
// looks at 6 bits) -- the same bits as "mant << (e - 88) or mant >> (88 - e)", in a dozen instructions instead of
// five nested exec-mask regions per value (phase B spent 57 % of its time issuing VALU work, most of it here).
__device__ __forceinline__ unsigned long long pb_to_fixed(float v, unsigned &bad) {
  const unsigned bits = __float_as_uint(v);
  const bool ok = bits <= 0x3F800000u;  // +0 .. 1.0; negative, > 1, inf and nan have larger bit patterns
  bad |= ok ? 0u : 1u;
  const unsigned e = bits >> 23;  // biased exponent (<= 127 when ok)
  const unsigned mant = (bits & 0x7FFFFFu) | 0x800000u;
  const unsigned long long m62 = (unsigned long long)(mant << 7) << 32;  // mant * 2^39
  unsigned sh = 127u - e;
  sh = sh > 63u ? 63u : sh;  // zero / denormal / tiny: everything is shifted out (m62 < 2^63)
  const unsigned long long r = m62 >> sh;
  return ok ? r : 0ull;
}

// The same conversion cut in two (PageRank): pb_encode runs ONCE PER SOURCE (phase A's slice, the tier tables) and
// leaves the value as (shift + 1) << 24 | 24-bit mantissa -- the same information as the fp32 -- and pb_decode, which
// phase B runs once PER EDGE, is two 32-bit shifts and one 64-bit shift: pb_decode(pb_encode(v)) == pb_to_fixed(v)
// bit for bit.  Phase B is bound by its per-edge VALU work, not by memory.
__device__ __forceinline__ uint32_t pb_encode(float v, unsigned &bad) {
  const unsigned bits = __float_as_uint(v);
  const bool ok = bits <= 0x3F800000u;
  bad |= ok ? 0u : 1u;
  const unsigned sh = 127u - (bits >> 23);  // wraps for e > 127, which is !ok
  const unsigned mant = (bits & 0x7FFFFFu) | 0x800000u;
  return (ok && sh < 63u) ? (((sh + 1u) << 24) | mant) : 0u;  // shifts >= 63 leave nothing of a 24-bit mantissa
}
__device__ __forceinline__ unsigned long long pb_decode(uint32_t enc) {
  return ((unsigned long long)(enc << 8) << 32) >> (enc >> 24);  // mantissa * 2^40 >> (shift + 1)
}

// phase A: vals[group(G[g]) + i] = x[chunk*CH + U[...]]   (TAG: see pb_accumulate_kernel)
template <int TAG = 0>
static __global__ void __launch_bounds__(PB_THREADS)
pb_expand_kernel(const float *__restrict__ x, int32_t m_global, int log_chunk, const eoff_t *__restrict__ chunk_ptr,
                 const uint32_t *__restrict__ chunk_order, const uint16_t *__restrict__ U,
                 const uint32_t *__restrict__ G, float *__restrict__ vals, const uint32_t *__restrict__ src_bits,
                 const uint32_t *__restrict__ chunk_lo, unsigned split, int log_group, int nt_store = 0,
                 // hub-ROW tier (nullable): the chunk's edges into the hr_n highest in-degree rows are not written to
                 // vals; their values are added up here, in hr_n u64 accumulators behind the slice, and ONE partial sum
                 // per (chunk, hub row) goes to hr_partial[chunk * hr_n + row] (integer sums: exact, order independent)
                 const eoff_t *__restrict__ hr_ptr = nullptr, const uint16_t *__restrict__ hr_U = nullptr,
                 const uint16_t *__restrict__ hr_R = nullptr, unsigned hr_n = 0,
                 unsigned long long *__restrict__ hr_partial = nullptr, unsigned *__restrict__ errflag = nullptr,
                 unsigned pad_slot = 0,  // PbPlan::chunk_slots (0 = 2^log_chunk)
                 PbTierRefresh tiers = PbTierRefresh()) {
  if (tiers.skip && *tiers.skip) return;
  extern __shared__ __attribute__((aligned(16))) float s_x[];
  // every workgroup first refreshes its share of the tier tables phase B reads (a few hundred entries: three tiny
  // launches between A and B otherwise)
  for (int t = 0; t < tiers.ntiers; t++) {
    const unsigned per = (tiers.slots[t] + gridDim.x - 1) / gridDim.x;
    const unsigned k0 = blockIdx.x * per, k1 = k0 + per < tiers.slots[t] ? k0 + per : tiers.slots[t];
    unsigned bad_t = 0u;
    for (unsigned k = k0 + threadIdx.x; k < k1; k += PB_THREADS)
      tiers.val[t][k] = k < tiers.n[t] ? __uint_as_float(pb_encode(x[tiers.ids[t][k]], bad_t)) : 0.0f;
    if (bad_t && errflag) *errflag = 1u;
  }
  __shared__ unsigned s_bits[PB_THREADS], s_pref[PB_THREADS], s_scr[PB_WAVES + 1];
  const unsigned ch = 1u << log_chunk;
  // `split` workgroups share one chunk (same LDS slice, consecutive parts of its edge range): more,
  // smaller work units so that the last round of workgroups does not leave most CUs idle
  const unsigned c = chunk_order[blockIdx.x / split];
  const unsigned part = blockIdx.x % split;
  const size_t base = (size_t)c << log_chunk;
  unsigned n_slots = ch;  // slots of the slice that hold a source value
  if (src_bits) {  // compacted slice: the chunk's active sources, gathered from their original id range
    if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0 && !(nt_store & 2))
      n_slots = pb_load_slice4(x, m_global, src_bits, chunk_lo[c], chunk_lo[c + 1], s_x, s_bits, s_pref, s_scr);
    else
      n_slots = pb_walk_slice(src_bits, chunk_lo[c], chunk_lo[c + 1], s_bits, s_pref, s_scr,
                              [&](unsigned id, unsigned k) { s_x[k] = x[id]; }, [](unsigned) {});
  } else if (base + ch <= (size_t)m_global) {  // whole slice in range: 16-byte loads
    const pb_f32x4 *x4 = reinterpret_cast<const pb_f32x4 *>(x + base);
    pb_f32x4 *s4 = reinterpret_cast<pb_f32x4 *>(s_x);
#pragma unroll 4
    for (unsigned i = threadIdx.x; i < (ch >> 2); i += PB_THREADS) s4[i] = x4[i];
  } else {
    for (unsigned i = threadIdx.x; i < ch; i += PB_THREADS) {
      const size_t g = base + i;
      s_x[i] = (g < (size_t)m_global) ? x[g] : 0.0f;
    }
  }
  const unsigned zslot = pad_slot ? pad_slot : ch;
  // the slice becomes fixed-point codes (pb_encode): the float -> fixed conversion is paid per source here, not per
  // edge in phase B; the code of 0.0 is 0, so the zero slot and the bits that travel through vals stay what they were
  __syncthreads();
  if (!(nt_store & 8)) {  // bit 3: RAW slice -- the 32-bit words of x travel as they are (integer sweeps, gdn_bc.hip)
    unsigned bad_a = 0u;
    uint32_t *s_u = reinterpret_cast<uint32_t *>(s_x);
    for (unsigned i = threadIdx.x; i < n_slots; i += PB_THREADS) s_u[i] = pb_encode(s_x[i], bad_a);
    if (bad_a && errflag) *errflag = 1u;
  }
  if (threadIdx.x == 0) s_x[zslot] = 0.0f;  // zero slot for pad edges
  unsigned long long *s_hr = reinterpret_cast<unsigned long long *>(s_x + zslot + 4);  // 16 bytes behind the zero slot
  // hub-row tier: PB_HR_THREADS threads (two waves) fold the hub-row edge list while the others run the main sweep --
  // the fold is a chain of short dependent steps (latency), the sweep is throughput, and both only READ the slice
  const bool with_hr = hr_ptr != nullptr && part == 0;
  if (with_hr)
    for (unsigned i = threadIdx.x; i < hr_n; i += PB_THREADS) s_hr[i] = 0ull;
  __syncthreads();
  const unsigned t0 = with_hr ? PB_HR_THREADS : 0u;      // first thread of the main sweep
  const unsigned nthr = PB_THREADS - t0;                  // its thread count
  // half-groups: lane pair (2i, 2i+1) handles group i; each lane loads 4 source ids (8 B),
  // gathers 4 values from LDS and stores 16 B, so a wave store covers whole 64-byte lines
  const eoff_t hc0 = chunk_ptr[c] >> 2, hc1 = chunk_ptr[c + 1] >> 2;
  const eoff_t hlen = (hc1 - hc0 + split - 1) / split;
  const eoff_t h0 = hc0 + (eoff_t)part * hlen;
  const eoff_t h1 = (h0 + hlen < hc1) ? h0 + hlen : hc1;
  const pb_u16x4 *U4 = reinterpret_cast<const pb_u16x4 *>(U);
  pb_f32x4 *X4 = reinterpret_cast<pb_f32x4 *>(vals);
  constexpr int UNR = 8;
  const int lq = log_group - 2;  // a lane owns a quad of 4 edges; 2^lq lanes share one entry of G
  const unsigned qmask = (1u << lq) - 1u;
  if (threadIdx.x >= t0) {
    for (eoff_t h = h0 + (threadIdx.x - t0); h < h1; h += (eoff_t)UNR * nthr) {
      pb_u16x4 u[UNR];
      unsigned d[UNR];
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const eoff_t hh = h + (eoff_t)r * nthr;
        if (hh < h1) {
          u[r] = __builtin_nontemporal_load(U4 + hh);
          d[r] = __builtin_nontemporal_load(G + (hh >> lq));
        }
      }
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const eoff_t hh = h + (eoff_t)r * nthr;
        if (hh < h1) {
          pb_f32x4 o;
          o.x = s_x[u[r].x];
          o.y = s_x[u[r].y];
          o.z = s_x[u[r].z];
          o.w = s_x[u[r].w];
          pb_f32x4 *dst = X4 + (((size_t)d[r]) << lq) + (size_t)((unsigned)hh & qmask);
          if (nt_store & 1) __builtin_nontemporal_store(o, dst);
          else *dst = o;
        }
      }
    }
  } else {
    const unsigned nq = (unsigned)((hr_ptr[c + 1] - hr_ptr[c]) >> 2);
    const pb_u16x4 *HU = reinterpret_cast<const pb_u16x4 *>(hr_U) + (hr_ptr[c] >> 2);
    const pb_u16x4 *HR = reinterpret_cast<const pb_u16x4 *>(hr_R) + (hr_ptr[c] >> 2);
    unsigned bad = 0u;
    // every thread folds ONE contiguous block of the (row-sorted) edge list: a hub row owns hundreds of consecutive
    // edges per chunk, and one atomic per lane and quad made its accumulator the bottleneck (phase A 2.2 -> 2.9 ms);
    // a block of ~20 quads turns that into one atomic per thread and run
    const unsigned blk = (nq + PB_HR_THREADS - 1) / PB_HR_THREADS;
    const unsigned qb = threadIdx.x * blk < nq ? threadIdx.x * blk : nq, qe = qb + blk < nq ? qb + blk : nq;
    unsigned cur = 0xFFFFFFFFu;
    unsigned long long acc = 0ull;
    constexpr int HRB = 8;  // quads loaded ahead of the fold
    for (unsigned q = qb; q < qe; q += HRB) {
      pb_u16x4 u[HRB], v[HRB];
#pragma unroll
      for (int j = 0; j < HRB; j++)
        if (q + j < qe) {
          u[j] = HU[q + j];
          v[j] = HR[q + j];
        }
#pragma unroll
      for (int j = 0; j < HRB; j++) {
        if (q + j < qe) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const unsigned row = v[j][k];
            const unsigned long long f = pb_decode(__float_as_uint(s_x[u[j][k]]));
            if (row == cur) acc += f;
            else {
              if (cur != 0xFFFFFFFFu) atomicAdd(&s_hr[cur], acc);
              cur = row;
              acc = f;
            }
          }
        }
      }
    }
    if (cur != 0xFFFFFFFFu) atomicAdd(&s_hr[cur], acc);
    if (bad) *errflag = 1u;
  }
  if (with_hr) {
    __syncthreads();
    for (unsigned i = threadIdx.x; i < hr_n; i += PB_THREADS) hr_partial[(size_t)c * hr_n + i] = s_hr[i];
  }
}

// hub-row totals: total[r] = SUM over chunks of partial[c * n + r].  64 rows x 16 chunk lanes per workgroup (a thread
// per row walking all chunks alone is one long chain of dependent-latency loads: 0.5 ms for 1792 chunks)
static __global__ void __launch_bounds__(PB_THREADS)
pb_hubrow_reduce_kernel(const unsigned long long *__restrict__ partial, unsigned nchunks, unsigned n,
                        unsigned long long *__restrict__ total) {
  __shared__ unsigned long long s_part[16][64];
  const unsigned rl = threadIdx.x & 63u, cl = threadIdx.x >> 6;
  const unsigned r = blockIdx.x * 64u + rl;
  unsigned long long t = 0;
  if (r < n)
    for (unsigned c = cl; c < nchunks; c += 16u) t += partial[(size_t)c * n + r];
  s_part[cl][rl] = t;
  __syncthreads();
  if (cl == 0 && r < n) {
#pragma unroll
    for (int j = 1; j < 16; j++) t += s_part[j][rl];
    total[r] = t;
  }
}

// phase A with a per-edge factor (SpMV): vals[8*G[g] + i] = A[8*g + i] * x[chunk*CH + U[8*g + i]]; A == nullptr: the
// pattern matrix (factor 1, no value stream)
static __global__ void __launch_bounds__(PB_THREADS)
pb_expand_scaled_kernel(const float *__restrict__ x, int32_t m_global, int log_chunk,
                        const eoff_t *__restrict__ chunk_ptr, const uint32_t *__restrict__ chunk_order,
                        const uint16_t *__restrict__ U, const uint32_t *__restrict__ G, const float *__restrict__ A,
                        float *__restrict__ vals, int log_group, const uint32_t *__restrict__ src_bits = nullptr,
                        const uint32_t *__restrict__ chunk_lo = nullptr, unsigned pad_slot = 0,
                        // nullable: max |x| over the values this launch loads, as float bits (atomicMax): the fixed-point
                        // scale of the accumulate phase needs it, and a pass of its own over x cost 10 % of a multiply
                        unsigned *__restrict__ absmax = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float s_x[];
  __shared__ unsigned s_bits[PB_THREADS], s_pref[PB_THREADS], s_scr[PB_WAVES + 1];
  const unsigned ch = 1u << log_chunk;
  const unsigned c = chunk_order[blockIdx.x];
  const size_t base = (size_t)c << log_chunk;
  unsigned n_slots = ch;
  if (src_bits) {  // compacted slice: the chunk's active columns, gathered from their original id range
    if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0)
      n_slots = pb_load_slice4(x, m_global, src_bits, chunk_lo[c], chunk_lo[c + 1], s_x, s_bits, s_pref, s_scr);
    else
      n_slots = pb_walk_slice(src_bits, chunk_lo[c], chunk_lo[c + 1], s_bits, s_pref, s_scr,
                              [&](unsigned id, unsigned k) { s_x[k] = x[id]; }, [](unsigned) {});
  } else {
    for (unsigned i = threadIdx.x; i < ch; i += PB_THREADS) {
      const size_t g = base + i;
      s_x[i] = (g < (size_t)m_global) ? x[g] : 0.0f;
    }
  }
  if (threadIdx.x == 0) s_x[pad_slot ? pad_slot : ch] = 0.0f;
  __syncthreads();
  if (absmax) {
    unsigned mx = 0u;
    for (unsigned i = threadIdx.x; i < n_slots; i += PB_THREADS) {
      const unsigned bts = __float_as_uint(s_x[i]) & 0x7FFFFFFFu;
      mx = (bts > mx && bts < 0x7F800000u) ? bts : mx;  // finite values only (see pb_tier_gather_f32_kernel)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned t = (unsigned)__shfl_xor((int)mx, o, 64);
      mx = t > mx ? t : mx;
    }
    // ONE atomic per workgroup: a hot address takes ~12 ns per atomic whoever issues it
    if (gdn_lane() == 0) s_scr[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned t = 0u;
      for (int w = 0; w < PB_WAVES; w++) t = s_scr[w] > t ? s_scr[w] : t;
      if (t) atomicMax(absmax, t);
    }
  }
  const eoff_t h0 = chunk_ptr[c] >> 2, h1 = chunk_ptr[c + 1] >> 2;
  const pb_u16x4 *U4 = reinterpret_cast<const pb_u16x4 *>(U);
  const pb_f32x4 *A4 = reinterpret_cast<const pb_f32x4 *>(A);
  pb_f32x4 *X4 = reinterpret_cast<pb_f32x4 *>(vals);
  constexpr int UNR = 4;
  const int lq = log_group - 2;
  const unsigned qmask = (1u << lq) - 1u;
  for (eoff_t h = h0 + threadIdx.x; h < h1; h += UNR * PB_THREADS) {
    pb_u16x4 u[UNR];
    pb_f32x4 a[UNR];
    unsigned d[UNR];
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t hh = h + (eoff_t)r * PB_THREADS;
      if (hh < h1) {
        u[r] = __builtin_nontemporal_load(U4 + hh);
        if (A) a[r] = __builtin_nontemporal_load(A4 + hh);
        d[r] = __builtin_nontemporal_load(G + (hh >> lq));
      }
    }
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t hh = h + (eoff_t)r * PB_THREADS;
      if (hh < h1) {
        pb_f32x4 o;
        o.x = s_x[u[r].x];
        o.y = s_x[u[r].y];
        o.z = s_x[u[r].z];
        o.w = s_x[u[r].w];
        if (A) {
          o.x = gdn_fmul(o.x, a[r].x);
          o.y = gdn_fmul(o.y, a[r].y);
          o.z = gdn_fmul(o.z, a[r].z);
          o.w = gdn_fmul(o.w, a[r].w);
        }
        X4[(((size_t)d[r]) << lq) + (size_t)((unsigned)hh & qmask)] = o;
      }
    }
  }
}

// signed value * 2^shift -> two's complement fixed point (SpMV); |v * 2^shift| must stay < 2^62.  Branch-free form
// of (long long)(v * scale) (truncation toward zero): mantissa at bits 39..62, one clamped right shift, conditional
// negation -- the compiler's float -> int64 conversion is a long divergent sequence, and phase B pays it per edge.
// *lossy (nullable): 0 < |v * 2^shift| < 2^23, i.e. mantissa bits may have been shifted out
__device__ __forceinline__ unsigned long long pb_to_fixed_signed(float v, float scale, unsigned &bad, bool *lossy = nullptr) {
  const float t = v * scale;  // exact: scale is a power of two
  const unsigned bits = __float_as_uint(t);
  const unsigned mag_bits = bits & 0x7FFFFFFFu;
  const bool ok = mag_bits < 0x5E800000u;  // |t| < 2^62 (also rejects inf / nan)
  bad |= ok ? 0u : 1u;
  const unsigned e = mag_bits >> 23;
  if (lossy) *lossy = mag_bits != 0u && e < 150u;
  const unsigned mant = (mag_bits & 0x7FFFFFu) | 0x800000u;
  const unsigned long long m62 = (unsigned long long)(mant << 7) << 32;  // mant * 2^39
  unsigned sh = 189u - e;  // |t| = mant * 2^(e - 150) = m62 >> (189 - e); e <= 188 when ok
  sh = sh > 63u ? 63u : sh;
  const unsigned long long mag = m62 >> sh;
  const unsigned long long r = (bits >> 31) ? (0ull - mag) : mag;
  return ok ? r : 0ull;
}

// Ops with `static constexpr bool kTrackLossy = true` (SpMV) get a per-bin LDS bitmap of the rows that received a product
// whose fixed-point conversion dropped bits (Op::lossy(v)); such a row whose sum is too small for the dropped bits not to
// matter is handed to the op as the PB_REPAIR_ROW pattern and recomputed exactly by it (gdn_spmv.hip).  A second bitmap
// marks the rows that took a product fixed point cannot hold at all (Op::must_repair(v): inf, nan, or beyond the
// scale): those are always recomputed, so they come out inf / nan exactly like the reference's fp32 loop
// (src/spmv/omp_base.cc:22-33).  Other ops: no code at all.
#define PB_REPAIR_ROW 0x7FC0DEADu  // a quiet NaN no arithmetic produces: from_fixed never returns NaN at all
template <class T, class = void>
struct PbTracksLossy {
  static constexpr bool value = false;
};
template <class T>
struct PbTracksLossy<T, decltype((void)T::kTrackLossy)> {
  static constexpr bool value = T::kTrackLossy;
};
#define PB_LOSSY_MIN_SUM (1ull << 32)  // units: below this a row with a lossy product is recomputed

// The timing-only ablations of phase B (GDN_PB_DBG bits: 1 no LDS atomics, 2 no epilogue, 4 scalar epilogue, 8 no record
// tiers, 32 no main stream -- WRONG results by construction) exist in GDN_EXPERIMENTS builds only (make EXPERIMENTS=1,
// tools/build_variant.sh): the shipped kernel has no such branch, the `dbg` argument is dead there.
#ifdef GDN_EXPERIMENTS
#define PB_DBG(bits) ((dbg & (bits)) != 0)
#else
#define PB_DBG(bits) false
#endif
// The whole 256-record blocks of every bin's record stream (ptr: nbins + 1 offsets, multiples of 4 records), and of the
// per-record factors A (nullable), lane-interleaved in place for phase B's form 2: word l of a block = records l, 64 + l,
// 128 + l, 192 + l.  A workgroup per bin, a wave per block (every load of the wave has returned before its stores issue).
static __global__ void __launch_bounds__(GDN_BLOCK)
pb_stream_interleave_kernel(uint32_t *__restrict__ rec, float *__restrict__ A, const eoff_t *__restrict__ ptr) {
  typedef unsigned pbi_u32x4 __attribute__((ext_vector_type(4)));
  const unsigned lane = gdn_lane();
  const eoff_t j0 = ptr[blockIdx.x];
  const unsigned nblk = (unsigned)((ptr[blockIdx.x + 1] - j0) >> 8);
  for (unsigned q = threadIdx.x >> 6; q < nblk; q += GDN_WAVES_PER_BLOCK) {
    uint32_t *base = rec + j0 + (eoff_t)q * 256u;
    pbi_u32x4 v = {base[lane], base[64u + lane], base[128u + lane], base[192u + lane]};
    pb_f32x4 fa = {0.0f, 0.0f, 0.0f, 0.0f};
    float *ab = A ? A + j0 + (eoff_t)q * 256u : nullptr;
    if (ab) fa = pb_f32x4{ab[lane], ab[64u + lane], ab[128u + lane], ab[192u + lane]};
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    reinterpret_cast<pbi_u32x4 *>(base)[lane] = v;
    if (ab) reinterpret_cast<pb_f32x4 *>(ab)[lane] = fa;
  }
}
// V of a plan whose bins all start on multiples of 512 edges, in place: inside every block of 512 edges (128 quads of four
// 16-bit rows) the 16-byte word l holds quad l and quad 64 + l (PbPlan::v_il; read by pb_accumulate_kernel's load_step).
// A wave per block.
static __global__ void __launch_bounds__(GDN_BLOCK)
pb_v_interleave_kernel(uint16_t *__restrict__ V, unsigned long long nblk) {
  typedef unsigned pbi_u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned pbi_u32x4 __attribute__((ext_vector_type(4)));
  const unsigned lane = gdn_lane();
  for (unsigned long long q = (unsigned long long)blockIdx.x * GDN_WAVES_PER_BLOCK + (threadIdx.x >> 6); q < nblk;
       q += (unsigned long long)gridDim.x * GDN_WAVES_PER_BLOCK) {
    pbi_u32x2 *base = reinterpret_cast<pbi_u32x2 *>(V + q * 512ull);
    const pbi_u32x2 lo = base[lane], hi = base[64u + lane];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    reinterpret_cast<pbi_u32x4 *>(base)[lane] = pbi_u32x4{lo.x, lo.y, hi.x, hi.y};
  }
}
// host side: interleaves p.V when every bin of the plan starts on a multiple of 512 edges (sets p.v_il); else leaves it
static inline int pb_v_interleave(PbPlan &p) {
  p.v_il = false;
  if (p.nbins == 0 || p.n_pad < 512) return GDN_OK;
  std::vector<eoff_t> bp((size_t)p.nbins + 1);
  GDN_HIP(hipMemcpy(bp.data(), p.bin_ptr.p, bp.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
  for (eoff_t x : bp)
    if (x & 511u) return GDN_OK;
  const unsigned long long nblk = (unsigned long long)bp[p.nbins] >> 9;
  if (nblk == 0) return GDN_OK;
  const unsigned long long wb = (nblk + GDN_WAVES_PER_BLOCK - 1) / GDN_WAVES_PER_BLOCK;
  hipLaunchKernelGGL(pb_v_interleave_kernel, dim3((unsigned)(wb > 65536ull ? 65536ull : wb)), dim3(GDN_BLOCK), 0, 0, p.V.p, nblk);
  GDN_HIP(hipGetLastError());
  p.v_il = true;
  return GDN_OK;
}

// phase B: acc[bin] = SUM fixed(vals) over the bin's contiguous range; then the fused epilogue of the rows (op).
// TAG: 1 = the launches of a plan's placement search (PbPlacer) -- the same code under another name, so that a profiler's
// per-kernel statistics of the iterations proper do not average in sweeps timed on allocations that were then dropped
template <class Op, int TAG = 0>
__global__ void __launch_bounds__(PB_THREADS)
pb_accumulate_kernel(int32_t m_local, int log_bin, const eoff_t *__restrict__ bin_ptr,
                     const uint32_t *__restrict__ bin_order, const uint16_t *__restrict__ V,
                     const float *__restrict__ vals, double *__restrict__ partial, unsigned *__restrict__ errflag,
                     const uint32_t *__restrict__ dst_bits, const uint32_t *__restrict__ bin_lo, Op op,
                     int dbg = 0,  // GDN_EXPERIMENTS builds only (PB_DBG above); ignored otherwise
                     unsigned bin_begin = 0,  // bin_order == nullptr: bins bin_begin + blockIdx.x (partial launches)
                     // hub tier (nullable): per bin a second stream of (hub index, row) pairs, sorted by hub, whose
                     // values are read from the small per-iteration table hub_val instead of travelling through vals
                     const eoff_t *__restrict__ hub_ptr = nullptr, const uint16_t *__restrict__ hub_U = nullptr,
                     const uint16_t *__restrict__ hub_V = nullptr, const float *__restrict__ hub_val = nullptr,
                     // delta-coded rows (PbPlan::v8, nullable): V is then not read
                     const uint8_t *__restrict__ Vd = nullptr, const uint16_t *__restrict__ Vb = nullptr,
                     // per-edge factor of the hub stream (SpMV's Ax in hub order, nullable): value = hub_val * hub_A
                     const float *__restrict__ hub_A = nullptr,
                     // hub-ROW tier (nullable): rows hrb_vl[hrb_ptr[b] .. hrb_ptr[b+1]) of this bin start from the
                     // totals phase A already summed (pb_expand_kernel), not from zero
                     const unsigned *__restrict__ hrb_ptr = nullptr, const uint16_t *__restrict__ hrb_vl = nullptr,
                     const unsigned long long *__restrict__ hr_total = nullptr,
                     // mid tiers: per bin one stream of (source index, row) records per tier, values from the tier's table
                     PbMidArgs mid = PbMidArgs(),
                     // tickets (nullable; PbParts above): parts by launch index
                     unsigned *__restrict__ ticket = nullptr, PbParts parts = PbParts()) {
  if (gdn_skip_launch(op)) return;
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_acc[];
  __shared__ double s_red[PB_WAVES];
  __shared__ unsigned s_bits[PB_THREADS], s_pref[PB_THREADS], s_scr[PB_WAVES + 1];
  const unsigned bn = 1u << log_bin;
  const unsigned b = bin_order ? bin_order[blockIdx.x] : bin_begin + blockIdx.x;
  for (unsigned i = threadIdx.x; i < bn; i += PB_THREADS) s_acc[i] = 0ull;
  constexpr bool LZ = PbTracksLossy<Op>::value;
  unsigned *s_lossy = reinterpret_cast<unsigned *>(s_acc + bn);  // bn / 32 words behind the accumulators (LZ launches only)
  unsigned *s_force = s_lossy + (bn >> 5);                        // and bn / 32 more: rows that must be recomputed
  if constexpr (LZ)
    for (unsigned i = threadIdx.x; i < (bn >> 4); i += PB_THREADS) s_lossy[i] = 0u;
  __syncthreads();
  if (hrb_ptr) {
    for (unsigned i = hrb_ptr[b] + threadIdx.x; i < hrb_ptr[b + 1]; i += PB_THREADS) s_acc[hrb_vl[i]] = hr_total[i];
    __syncthreads();
  }
  const unsigned lane = gdn_lane();
  const unsigned w = threadIdx.x >> 6;
  const eoff_t q0 = bin_ptr[b] >> 2, q1 = bin_ptr[b + 1] >> 2;  // units of 4 edges
  const pb_f32x4 *X4 = reinterpret_cast<const pb_f32x4 *>(vals);
  const pb_u16x4 *V4 = reinterpret_cast<const pb_u16x4 *>(V);
  unsigned bad = 0u;
  // value -> fixed point for `row` of this bin (+ the lossy mark of the ops that track it)
  // fxl: the conversion + whether it dropped bits; mark(row): the lossy mark.  Call sites convert a group of values
  // first and branch ONCE on "any of them lossy" (one exec-mask region per group instead of one per product: the
  // tracking cost 7 % of an SpMV product by product)
  auto fxl = [&](float val, bool &lz) -> unsigned long long {
    if constexpr (LZ) {
      return op.to_fixed_lossy(val, bad, lz);
    } else {
      lz = false;
      return op.to_fixed(val, bad);
    }
  };
  auto mark = [&](unsigned row, float val) {
    atomicOr(&s_lossy[row >> 5], 1u << (row & 31u));
    if constexpr (LZ) {
      if (op.must_repair(val)) atomicOr(&s_force[row >> 5], 1u << (row & 31u));
    }
  };
  auto fx = [&](float val, unsigned row) -> unsigned long long {
    bool lz;
    const unsigned long long f = fxl(val, lz);
    if constexpr (LZ) {
      if (lz) mark(row, val);
    }
    return f;
  };
  constexpr int UNR = 4;
  pb_f32x4 xs[UNR], nx[UNR];
  pb_u16x4 vs[UNR], nv[UNR];
  // The bin's range as 32-bit quad indices behind wave-uniform base pointers (scalar base + 32-bit lane offset
  // addressing instead of 64-bit pointer arithmetic per load), and the steps in which every lane has work run
  // without bounds predicates: phase B has no spare issue slots (one 1024-thread workgroup per CU).
  const unsigned nq = (unsigned)(q1 - q0);
  const pb_f32x4 *Xq = X4 + q0;
  const pb_u16x4 *Vq = V4 + q0;
  const uint32_t *Dq = reinterpret_cast<const uint32_t *>(Vd) + q0;  // delta-coded rows: 4 distance bytes per quad
  const uint16_t *Bq = Vb + (q0 >> 3);                                //   + the base row of every 32-edge group
  // the 4 rows of quad i.  Delta-coded stream: the 8 lanes of a group (bin ranges are whole groups, so a group is
  // valid or not as a whole) scan their quad sums.  Called by whole 8-lane groups.
  auto load_rows = [&](unsigned i, bool ok) -> pb_u16x4 {
    pb_u16x4 o = {0, 0, 0, 0};
    if (!Vd) {
      if (ok) o = __builtin_nontemporal_load(Vq + i);
      return o;
    }
    unsigned d = 0, base = 0;
    if (ok) {
      d = __builtin_nontemporal_load(Dq + i);
      base = __builtin_nontemporal_load(Bq + (i >> 3));
    }
    const unsigned d0 = d & 255u, d1 = (d >> 8) & 255u, d2 = (d >> 16) & 255u, d3 = d >> 24;
    const unsigned tot = d0 + d1 + d2 + d3;
    unsigned incl = tot;
    const unsigned l8 = lane & 7u;
    // DPP row_shr:k (0x110 + k) stays in the VALU; a lane with l8 >= k reads a lane of its own group
    unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);
    if (l8 >= 1u) incl += t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);
    if (l8 >= 2u) incl += t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);
    if (l8 >= 4u) incl += t;
    const unsigned r0 = base + incl - tot + d0;
    o.x = (unsigned short)r0;
    o.y = (unsigned short)(r0 + d1);
    o.z = (unsigned short)(r0 + d1 + d2);
    o.w = (unsigned short)(r0 + d1 + d2 + d3);
    return o;
  };
  // edges of one tile are sorted by destination row: fold equal neighbours in the lane first (a hub row receives
  // hundreds of consecutive edges per tile)
  auto fold = [&](const pb_f32x4 &x, const pb_u16x4 &v) {
    if (PB_DBG(1)) {
      bad |= (unsigned)(x.x + x.y + x.z + x.w == 123.456f) + (unsigned)(v.x + v.w == 77777u);
      return;
    }
    // run sums by selects, then one predicated atomic per run end (no nested divergent regions)
    bool l0, l1, l2, l3;
    const unsigned long long f0 = fxl(x.x, l0), f1 = fxl(x.y, l1), f2 = fxl(x.z, l2), f3 = fxl(x.w, l3);
    if constexpr (LZ) {
      if (l0 | l1 | l2 | l3) {
        if (l0) mark(v.x, x.x);
        if (l1) mark(v.y, x.y);
        if (l2) mark(v.z, x.z);
        if (l3) mark(v.w, x.w);
      }
    }
    const bool e1 = v.y == v.x, e2 = v.z == v.y, e3 = v.w == v.z;
    const unsigned long long p1 = f1 + (e1 ? f0 : 0ull);
    const unsigned long long p2 = f2 + (e2 ? p1 : 0ull);
    const unsigned long long p3 = f3 + (e3 ? p2 : 0ull);
    if (!e1) atomicAdd(&s_acc[v.x], f0);
    if (!e2) atomicAdd(&s_acc[v.y], p1);
    if (!e3) atomicAdd(&s_acc[v.z], p2);
    atomicAdd(&s_acc[v.w], p3);
  };
  constexpr unsigned STEPU = (unsigned)UNR * PB_THREADS;
  static_assert(UNR % 2 == 0, "the interleaved-V form pairs the quads of a step");
  // Quad r of this thread in the step that starts at quad `sbase`.  Plain V: threadIdx.x + r * PB_THREADS.  Interleaved V
  // (mid.v_il; the bin then starts and ends on multiples of 128 quads): a wave owns 128 consecutive quads per pair of r, lane l
  // the quads l and 64 + l of them -- whose rows are ONE 16-byte word of V (pb_v_interleave_kernel): three vector-memory
  // loads per 8 edges instead of four.
  const bool vil = mid.v_il != 0;
  auto qi = [&](unsigned sbase, int r) -> unsigned {
    return vil ? sbase + (unsigned)(r >> 1) * (2u * PB_THREADS) + (w << 7) + ((unsigned)(r & 1) << 6) + lane
               : sbase + threadIdx.x + (unsigned)r * PB_THREADS;
  };
  const pb_u16x8 *V8q = reinterpret_cast<const pb_u16x8 *>(Vq);
  auto load_step = [&](unsigned sbase, pb_f32x4 (&x)[UNR], pb_u16x4 (&v)[UNR], bool full) {
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const unsigned i = qi(sbase, r);
      if (full || i < nq) x[r] = __builtin_nontemporal_load(Xq + i);
    }
    if (vil) {
#pragma unroll
      for (int pr = 0; pr < UNR / 2; pr++) {
        const unsigned i0 = qi(sbase, 2 * pr);  // (valid together with i0 + 64: nq is a multiple of 128)
        pb_u16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
        if (full || i0 < nq) t = __builtin_nontemporal_load(V8q + (((i0 - lane) >> 1) + lane));
        v[2 * pr] = pb_u16x4{t[0], t[1], t[2], t[3]};
        v[2 * pr + 1] = pb_u16x4{t[4], t[5], t[6], t[7]};
      }
    } else {
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const unsigned i = qi(sbase, r);
        v[r] = load_rows(i, full || i < nq);
      }
    }
  };
  unsigned sb = 0;  // first quad of the current step (wave-uniform)
  if (PB_DBG(32)) sb = nq;  // timing-only ablation: no main stream
  else if (nq >= STEPU) {
    // software pipeline over the full steps: the loads of step k+1 are in flight while step k is folded into LDS
    load_step(0u, xs, vs, true);
    for (; sb + 2u * STEPU <= nq; sb += STEPU) {
      load_step(sb + STEPU, nx, nv, true);
#pragma unroll
      for (int r = 0; r < UNR; r++) fold(xs[r], vs[r]);
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        xs[r] = nx[r];
        vs[r] = nv[r];
      }
    }
#pragma unroll
    for (int r = 0; r < UNR; r++) fold(xs[r], vs[r]);
    sb += STEPU;
  }
  // the last, partial step (q0 and q1 are multiples of 8 quads, so the 8 lanes of a group are valid together)
  load_step(sb, xs, vs, false);
#pragma unroll
  for (int r = 0; r < UNR; r++) {
    if (qi(sb, r) < nq) fold(xs[r], vs[r]);
  }
  if (hub_ptr && !PB_DBG(16)) {
    // edges of hub sources: 4 B per edge (u16 hub index + u16 row), 8 edges per lane and step with 16-byte loads;
    // the values come from a table that stays in L2: one 4-byte gather per DISTINCT hub of the lane's run (the
    // stream is sorted by hub, so a run of 8 edges holds 1-4 hubs)
    const unsigned nh = (unsigned)((hub_ptr[b + 1] - hub_ptr[b]) >> 3);  // octets (bin ranges are multiples of 16)
    const pb_u16x8 *HU = reinterpret_cast<const pb_u16x8 *>(hub_U) + (hub_ptr[b] >> 3);
    const pb_u16x8 *HV = reinterpret_cast<const pb_u16x8 *>(hub_V) + (hub_ptr[b] >> 3);
    constexpr int HUNR = 2;
    for (unsigned o0 = threadIdx.x; o0 < nh; o0 += (unsigned)HUNR * PB_THREADS) {
      pb_u16x8 hu[HUNR], hv[HUNR];
      float f0[HUNR];
#pragma unroll
      for (int r = 0; r < HUNR; r++) {
        const unsigned o = o0 + (unsigned)r * PB_THREADS;
        if (o < nh) {
          hu[r] = __builtin_nontemporal_load(HU + o);
          hv[r] = __builtin_nontemporal_load(HV + o);
        }
      }
#pragma unroll
      for (int r = 0; r < HUNR; r++) {
        const unsigned o = o0 + (unsigned)r * PB_THREADS;
        if (o < nh) f0[r] = hub_val[hu[r][0]];
      }
      if (hub_A) {  // every edge has its own factor: product rounded like the main path (phase A), then converted
        const pb_f32x4 *HA = reinterpret_cast<const pb_f32x4 *>(hub_A) + 2 * (hub_ptr[b] >> 3);
#pragma unroll
        for (int r = 0; r < HUNR; r++) {
          const unsigned o = o0 + (unsigned)r * PB_THREADS;
          if (o < nh) {
            const pb_f32x4 a0 = __builtin_nontemporal_load(HA + 2 * (size_t)o), a1 = __builtin_nontemporal_load(HA + 2 * (size_t)o + 1);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            float xv = f0[r];
#pragma unroll
            for (int k = 0; k < 8; k++) {
              if (k > 0 && hu[r][k] != hu[r][k - 1]) xv = hub_val[hu[r][k]];
              atomicAdd(&s_acc[hv[r][k]], fx(gdn_fmul(xv, av[k]), hv[r][k]));
            }
          }
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < HUNR; r++) {
        const unsigned o = o0 + (unsigned)r * PB_THREADS;
        if (o < nh) {
          unsigned long long a = op.to_fixed(f0[r], bad);
          atomicAdd(&s_acc[hv[r][0]], a);
#pragma unroll
          for (int k = 1; k < 8; k++) {
            if (hu[r][k] != hu[r][k - 1]) a = op.to_fixed(hub_val[hu[r][k]], bad);
            atomicAdd(&s_acc[hv[r][k]], a);
          }
        }
      }
    }
  }
  for (int t = 0; t < (PB_DBG(8) ? 0 : mid.n); t++) {
    // record streams, sorted by source.  Two forms (PbMidArgs::form):
    //  0  one record per lane and load (mid tiers): the 64 table reads of a wave instruction fall into a few
    //     consecutive lines -- near-coalesced L2 hits, not a divergent gather
    //  1  four consecutive records per lane (hubs: several records per source and bin): ONE 16-byte read of the table
    //     window [k0, k0 + 4) behind the lane's first source serves the four of them; a record further away (rare)
    //     falls back to its own read.  A per-lane table read costs vector-memory issue time whatever it hits, and
    //     this form needs a quarter of them.
    const float *__restrict__ T = mid.val[t];
    const float *__restrict__ FA = mid.A[t] ? mid.A[t] + mid.ptr[t][b] : nullptr;  // per-record factors
    const unsigned z = mid.zrec[t];
    constexpr unsigned RMASK = (1u << PB_MID_ROW_BITS) - 1u;
    typedef unsigned pb_u32x4 __attribute__((ext_vector_type(4)));
    if (mid.form[t] == 0 || mid.form[t] == 2) {
      const uint32_t *__restrict__ R = mid.rec[t] + mid.ptr[t][b];
      const unsigned nr = (unsigned)(mid.ptr[t][b + 1] - mid.ptr[t][b]);
      // form 2: the stream's whole blocks of 256 records are lane-interleaved (pt_interleave_kernel / pb_stream_interleave_
      // kernel): quad l of a block = records l, 64 + l, 128 + l, 192 + l of the sorted stream, so a wave's 64 quads are one
      // block.  A quarter of form 0's record (and factor) loads, and table read j of the wave still covers 64 consecutive
      // records.  Per-record factors, when present, are stored in the same order.  The rest of the stream (< 256 records)
      // is plain and read as form 0.
      const unsigned nfull = mid.form[t] == 2 ? nr & ~255u : 0u;
      if (nfull) {
        const pb_u32x4 *__restrict__ R4 = reinterpret_cast<const pb_u32x4 *>(R);
        const pb_f32x4 *__restrict__ FA4 = reinterpret_cast<const pb_f32x4 *>(FA);
        const unsigned nq4 = nfull >> 2;
        constexpr int IU = PB_IL_QUADS;
        for (unsigned i0 = threadIdx.x; i0 < nq4; i0 += (unsigned)IU * PB_THREADS) {
          pb_u32x4 rc[IU];
          pb_f32x4 a[IU];
          float f[IU][4];
#pragma unroll
          for (int r = 0; r < IU; r++) {
            const unsigned i = i0 + (unsigned)r * PB_THREADS;
            rc[r] = pb_u32x4{z, z, z, z};
            a[r] = pb_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (i < nq4) {
              rc[r] = __builtin_nontemporal_load(R4 + i);
              if (FA4) a[r] = __builtin_nontemporal_load(FA4 + i);
            }
          }
#pragma unroll
          for (int r = 0; r < IU; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) f[r][j] = T[rc[r][j] >> PB_MID_ROW_BITS];
#pragma unroll
          for (int r = 0; r < IU; r++)
#pragma unroll
            for (int j = 0; j < 4; j++)
              atomicAdd(&s_acc[rc[r][j] & RMASK], fx(FA4 ? gdn_fmul(f[r][j], a[r][j]) : f[r][j], rc[r][j] & RMASK));
        }
      }
      constexpr int MUNR = 8;
      constexpr unsigned RSTEP = (unsigned)MUNR * PB_THREADS;
      for (unsigned i0 = nfull + threadIdx.x; i0 < nr; i0 += RSTEP) {
        uint32_t rc[MUNR];
        float f[MUNR], a[MUNR];
#pragma unroll
        for (int r = 0; r < MUNR; r++) {
          const unsigned i = i0 + (unsigned)r * PB_THREADS;
          rc[r] = z;
          a[r] = 0.0f;
          if (i < nr) {
            rc[r] = __builtin_nontemporal_load(R + i);
            if (FA) a[r] = __builtin_nontemporal_load(FA + i);
          }
        }
#pragma unroll
        for (int r = 0; r < MUNR; r++) f[r] = T[rc[r] >> PB_MID_ROW_BITS];
        bool lz[MUNR], any = false;
        unsigned long long fv[MUNR];
#pragma unroll
        for (int r = 0; r < MUNR; r++) {
          if (FA) f[r] = gdn_fmul(f[r], a[r]);
          fv[r] = fxl(f[r], lz[r]);
          any |= lz[r];
        }
        if constexpr (LZ) {
          if (any) {
#pragma unroll
            for (int r = 0; r < MUNR; r++)
              if (lz[r]) mark(rc[r] & RMASK, f[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < MUNR; r++) atomicAdd(&s_acc[rc[r] & RMASK], fv[r]);
      }
      continue;
    }
    typedef float pb_f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // 4-byte aligned window
    const pb_u32x4 *__restrict__ R4 = reinterpret_cast<const pb_u32x4 *>(mid.rec[t] + mid.ptr[t][b]);
    const pb_f32x4 *__restrict__ FA4 = reinterpret_cast<const pb_f32x4 *>(FA);
    const unsigned nr4 = (unsigned)((mid.ptr[t][b + 1] - mid.ptr[t][b]) >> 2);
    constexpr int MU = 4;
    for (unsigned i0 = threadIdx.x; i0 < nr4; i0 += (unsigned)MU * PB_THREADS) {
      pb_u32x4 rc[MU];
      pb_f32x4 w[MU], a[MU];
#pragma unroll
      for (int r = 0; r < MU; r++) {
        const unsigned i = i0 + (unsigned)r * PB_THREADS;
        rc[r] = pb_u32x4{z, z, z, z};
        a[r] = pb_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (i < nr4) {
          rc[r] = __builtin_nontemporal_load(R4 + i);
          if (FA4) a[r] = __builtin_nontemporal_load(FA4 + i);
        }
      }
#pragma unroll
      for (int r = 0; r < MU; r++) w[r] = *reinterpret_cast<const pb_f32x4_a4 *>(T + (rc[r].x >> PB_MID_ROW_BITS));
#pragma unroll
      for (int r = 0; r < MU; r++) {
        const unsigned k0 = rc[r].x >> PB_MID_ROW_BITS;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const unsigned d = (rc[r][j] >> PB_MID_ROW_BITS) - k0;
          float f = d == 0 ? w[r].x : (d == 1 ? w[r].y : (d == 2 ? w[r].z : w[r].w));
          if (d > 3u) f = T[rc[r][j] >> PB_MID_ROW_BITS];  // also a source in FRONT of k0 (never in a sorted stream)
          atomicAdd(&s_acc[rc[r][j] & RMASK], fx(FA4 ? gdn_fmul(f, a[r][j]) : f, rc[r][j] & RMASK));
        }
      }
    }
  }
  __syncthreads();
  double dsum = 0.0;
  if (!PB_DBG(2)) {
    // compacted bin: its original row range (rows without in-edges get sum 0); plain bin: its 2^log_bin rows
    const size_t row0 = (size_t)b << log_bin;
    const unsigned lo = dst_bits ? bin_lo[b] : (unsigned)row0;
    const unsigned hi = dst_bits ? bin_lo[b + 1] : (unsigned)((row0 + bn < (size_t)m_local) ? row0 + bn : (size_t)m_local);
    // the row's sum; PB_REPAIR_ROW = "recompute me" for a row that took a lossy product and whose sum is too small to
    // hide it, or a product fixed point cannot hold
    auto row_sum = [&](unsigned k) -> float {
      const unsigned long long a = s_acc[k];
      if constexpr (LZ) {
        if ((s_lossy[k >> 5] >> (k & 31u)) & 1u) {
          const unsigned long long mag = (long long)a < 0 ? 0ull - a : a;
          if (mag < PB_LOSSY_MIN_SUM || ((s_force[k >> 5] >> (k & 31u)) & 1u)) return __uint_as_float(PB_REPAIR_ROW);
        }
      }
      return op.from_fixed(a, bad);
    };
    if (op.vec_ok && !PB_DBG(4)) dsum = pb_epilogue4(dst_bits, lo, hi, s_bits, s_pref, s_scr, op, row_sum);
    else dsum = pb_epilogue(dst_bits, lo, hi, s_bits, s_pref, s_scr, op, row_sum);
  }
  if (bad) *errflag = 1u;
  dsum = gdn_wave_sum(dsum);
  if (lane == 0) s_red[w] = dsum;
  // (tickets: every wave's row stores have left the wave before the barrier -- MI355X_MICROARCH.md, producer form)
  if (ticket) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < PB_WAVES; i++) t += s_red[i];
    partial[b] = t;
    if (ticket) {
      // the workgroup's rows (plain stores, dirty in this XCD's L2) written back, THEN the count: the waiter's successor
      // is a kernel launch of its own (acquire at its start)
      if (parts.mode == 0u) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      unsigned j = 0;
      while (j + 1u < parts.n && blockIdx.x >= parts.end[j]) j++;
      __hip_atomic_fetch_add(ticket + PB_TICKET_STRIDE * j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// the waiter of a part's tickets: one lane polls (L2-served loads, asleep in between) until the counter has reached
// `target` (wrap-safe), for at most ~4 s of device time -- then it raises *timeout instead of hanging the queue behind it
static __global__ void pb_ticket_wait_kernel(const unsigned *ticket, unsigned target, unsigned *timeout) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  while ((int)(__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
    __builtin_amdgcn_s_sleep(32);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) {
      __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}
// the tickets of an iteration that ran as ONE kernel without them (merge-path layout): add[j] to counter j, behind it
static __global__ void pb_ticket_add_kernel(unsigned *ticket, PbParts add) {
  if (threadIdx.x < add.n) __hip_atomic_fetch_add(ticket + PB_TICKET_STRIDE * threadIdx.x, add.end[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif  // __HIPCC__
