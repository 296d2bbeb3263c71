// gdn_pbtier.hpp -- the propagation-blocked layout of gdn_pb.hpp AND its record tiers from ONE gather pass over the edges
// (round 4).  Included by gdn_build.hip (shares its histogram / sampling kernels); nothing here is on a solver's hot path.
//
// Why: pb_build (gdn_build.hip) makes one layout per call -- mark pass, key pass, global 8-byte-key radix sort -- and a
// PageRank plan is six layouts (main + hubs + four mid tiers): ~480 ps per edge, 50 passes over the edges, so the
// one-call drop-ins (gdn_pr = PRSolver, src/pr/main.cc:19) never got the blocked layout the reference builds inside its
// solver (src/pr/push_pb.cu:271, include/prop_blocking.h:29-65).  Here the work is what it has to be:
//
//   vertex phase   per source: class (0 main, 1 hubs, 2.. mid tiers) and index inside the class, all classes ranked by
//                  ONE two-level scan (block counts, scan of the block counts, in-block ranks); per row: compact index,
//                  bin, row-in-bin; a bitmap with one bit per edge marks the first edge of every row.
//                  smap[source id] = class << 29 | index -- the only table the edges gather from.
//   pt_keygen      ONE pass over the edges in CSR order, a workgroup per destination bin (bins are contiguous edge
//                  ranges of an in-CSR), 64 consecutive edges per wave step: code S[e] = smap[col[e]] (THE divergent
//                  gather of the build: ~18 ps per edge at RMAT-27, everything else streams), row-in-bin R[e] from the
//                  popcount of the row-start bitmap, tile counts (bin x chunk) and tier counts in LDS.
//   (scans: padded tile offsets in chunk-major and bin-major order, segment offsets, tier bin offsets)
//   pt_split       stable partition of every bin's edges by (chunk >> 3 | tier): a workgroup walks its bin in steps of
//                  8 K edges, ranks them per wave by ballot matching, puts the step in digit order in LDS and writes
//                  runs.  Main-layout edges become 32-bit items (slot, row, chunk & 7), tier edges their final records.
//   pt_tiles       a wave per (bin, 8 chunks) segment: stable split by the last 3 chunk bits straight into V (contiguous)
//                  and U (one run per tile).
//   pt_radix       the records of a (bin, tier) sorted by source: two stable passes of <= 10 bits with the same LDS staging
//                  (CSR order is (row, source); phase B wants (source, row) so that its table reads fall into few lines).
//
// Order inside a tile is (row, source) as pb_build makes it, inside a bin's record stream (source, row): the layout
// is deterministic, and no result depends on it anyway (integer accumulation, gdn_pb.hpp).
// Limits (the caller falls back to pb_build): <= 8192 source chunks, bins of <= 2^14 rows, in-CSR orientation.
#pragma once

#define PT_THREADS 1024                 // pt_keygen
#define PT_PTHREADS 512                 // the partition kernels: 256 VGPRs per lane (16 items + digits + ranks live at once:
#define PT_WAVES (PT_PTHREADS / 64)     //   1024 threads spilled 1 KB per lane), two workgroups per CU
#define PT_IPT 16
#define PT_STEP (PT_PTHREADS * PT_IPT)  // 8192 items per step of a partition
#define PT_MAX_DIGITS 1024
#define PT_LOW_BITS 3                  // chunk bits left to pt_tiles (slot 15 + row 14 + 3 = one 32-bit item)
#define PT_INACTIVE 0xFFFFFFFFu
#define PT_CLASS_SHIFT 29
#define PT_IDX_MASK 0x1FFFFFFFu
#define PT_MAX_CHUNKS 8192u
#define PT_VTILE 2048                  // vertices per workgroup of the vertex-phase kernels (256 threads x 8)
#define PT_NCLS (1 + PB_MAX_REC_TIERS) // class 0 + record tiers
static_assert(PT_NCLS <= 6, "pt_src_assign_kernel packs the class counts of a thread into two u64 of three 16-bit fields");

// ---- scratch: ONE block of the scratch pool per phase, bump-allocated (gdn_scratch_malloc, gdn_common.hpp)
struct PtArena {
  char *base = nullptr;
  size_t cap = 0, used = 0;
  int init(size_t bytes, int site = 1) {
    release();
    bytes += 4096;
    void *q = nullptr;
    GDN_TRY(gdn_scratch_malloc(&q, bytes, site));
    base = static_cast<char *>(q);
    cap = bytes;
    used = 0;
    return GDN_OK;
  }
  template <class T>
  T *get(size_t n) {
    const size_t at = (used + 255) & ~(size_t)255;
    const size_t need = (n ? n : 1) * sizeof(T);
    if (at + need > cap) return nullptr;
    used = at + need;
    return reinterpret_cast<T *>(base + at);
  }
  void release() {
    if (base) gdn_scratch_free(base);
    base = nullptr;
    cap = used = 0;
  }
  ~PtArena() { release(); }
};
static inline size_t pt_pad256(size_t bytes) { return (bytes + 511) & ~(size_t)255; }
// Zeroing inside an arena is done by a kernel, not by hipMemsetAsync: on stream-ordered (hipMallocAsync) memory this runtime
// runs such a memset BEHIND the kernel queued after it (tools/memset_probe.hip, profiles/r04_memset_probe.txt -- the builder
// lost its row-start bits that way).  The arenas are hipMalloc blocks of the scratch cache now; the kernel stays.
static inline int pt_zero(void *p, size_t bytes) {
  const unsigned long long nw = (bytes + 3) / 4;
  if (nw == 0) return GDN_OK;
  const unsigned long long nb = (nw + GDN_BLOCK - 1) / GDN_BLOCK;
  hipLaunchKernelGGL(pb_fill_u32_kernel, dim3((unsigned)(nb > 65536ull ? 65536ull : nb)), dim3(GDN_BLOCK), 0, 0, static_cast<uint32_t *>(p), nw, 0u);
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

// ---- device-wide scans without allocation or synchronisation: rows of u32 block counts scanned by one workgroup per row
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_scan_rows_kernel(uint32_t *__restrict__ data, unsigned n, uint32_t *__restrict__ totals) {
  __shared__ uint32_t s[GDN_WAVES_PER_BLOCK];
  uint32_t *row = data + (size_t)blockIdx.x * n;
  uint32_t carry = 0;
  for (unsigned base = 0; base < n; base += GDN_BLOCK) {
    const unsigned i = base + threadIdx.x;
    const uint32_t v = i < n ? row[i] : 0u;
    uint32_t total;
    const uint32_t ex = gdn_block_excl_scan(v, s, &total);
    if (i < n) row[i] = carry + ex;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// u32 -> u64 exclusive scan, out[n] = total; ws: ceil(n / 2048) + 1 u64.  Three launches on the null stream.
int gdn_exclusive_scan_u32_to_u64_ws(const uint32_t *d_in, eoff_t *d_out, size_t n, eoff_t *ws, hipStream_t s);

// ---- vertex phase ---------------------------------------------------------------------------------------------------
struct PtSrcArgs {
  const int32_t *src_count;  // exact out-edge counts (trusted superset of "occurs as a column"), or nullptr:
  const uint32_t *cnt16;     //   sampled counts (pb_hub_sample_kernel) ...
  const uint32_t *mark;      //   ... and exact marks
  unsigned thr[PB_MAX_REC_TIERS];  // descending; tier t: count >= thr[t] (units: edges with src_count, else 16 edges)
  int ntiers;
  unsigned n;  // sources
};
// class of source s: 0 main, 1 + t record tier t, PT_NCLS = inactive
__device__ __forceinline__ int pt_class_of(const PtSrcArgs &a, unsigned s) {
  unsigned c16;
  bool act;
  if (a.src_count) {
    const int32_t d = a.src_count[s];
    act = d > 0;
    c16 = act ? (unsigned)d : 0u;
  } else {
    act = a.mark[s] != 0u;
    c16 = a.cnt16[s];
  }
  if (!act) return PT_NCLS;
  int k = 0;
  for (int t = a.ntiers - 1; t >= 0; t--)
    if (c16 >= a.thr[t]) k = t + 1;
  return k;
}

// block counts per class: bc[k * nb + block]
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_src_count_kernel(PtSrcArgs a, unsigned nb, uint32_t *__restrict__ bc) {
  __shared__ unsigned s_n[PT_NCLS];
  if (threadIdx.x < PT_NCLS) s_n[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned base = blockIdx.x * PT_VTILE + threadIdx.x * 8u;
  unsigned cnt[PT_NCLS];
#pragma unroll
  for (int k = 0; k < PT_NCLS; k++) cnt[k] = 0u;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const unsigned s = base + (unsigned)i;
    if (s < a.n) {
      const int k = pt_class_of(a, s);
#pragma unroll
      for (int q = 0; q < PT_NCLS; q++) cnt[q] += (k == q) ? 1u : 0u;
    }
  }
#pragma unroll
  for (int k = 0; k < PT_NCLS; k++) {
    const unsigned t = gdn_wave_sum(cnt[k]);
    if (gdn_lane() == 0 && t) atomicAdd(&s_n[k], t);
  }
  __syncthreads();
  if (threadIdx.x < PT_NCLS) bc[(size_t)threadIdx.x * nb + blockIdx.x] = s_n[threadIdx.x];
}

// ranks: smap[s] = class << 29 | index (class 0: chunk << log_chunk | slot), the ascending id list of every tier, the
// activity bitmap of the main layout's sources and the first id of every chunk
struct PtSrcOut {
  uint32_t *smap;
  uint32_t *ids[PB_MAX_REC_TIERS];
  uint8_t *src_bits8;  // byte view of the bitmap (bit i of word w <-> id 32 w + i, little endian)
  uint32_t *chunk_lo;
  unsigned per_c;
  int log_chunk;
};
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_src_assign_kernel(PtSrcArgs a, unsigned nb, const uint32_t *__restrict__ bc, PtSrcOut o) {
  __shared__ unsigned long long s_scan[GDN_WAVES_PER_BLOCK];
  const unsigned base = blockIdx.x * PT_VTILE + threadIdx.x * 8u;
  int cls[8];
  // in-thread counts, packed 16 bits per class: A = classes 0..2, B = classes 3..5
  unsigned long long pa = 0ull, pb = 0ull;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const unsigned s = base + (unsigned)i;
    cls[i] = s < a.n ? pt_class_of(a, s) : PT_NCLS;
    if (cls[i] < 3) pa += 1ull << (16 * cls[i]);
    else if (cls[i] < PT_NCLS) pb += 1ull << (16 * (cls[i] - 3));
  }
  unsigned long long ta, tb;
  const unsigned long long ea = gdn_block_excl_scan(pa, s_scan, &ta);
  __syncthreads();
  const unsigned long long eb = gdn_block_excl_scan(pb, s_scan, &tb);
  unsigned run[PT_NCLS];
#pragma unroll
  for (int k = 0; k < PT_NCLS; k++) {
    const unsigned long long e = k < 3 ? ea : eb;
    run[k] = bc[(size_t)k * nb + blockIdx.x] + (unsigned)((e >> (16 * (k % 3))) & 0xFFFFull);
  }
  unsigned bits = 0u;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const unsigned s = base + (unsigned)i;
    if (s >= a.n) continue;
    const int k = cls[i];
    unsigned code = PT_INACTIVE;
    if (k == 0) {
      const unsigned r = run[0]++;
      const unsigned c = r / o.per_c, sl = r - c * o.per_c;
      code = (c << o.log_chunk) | sl;
      if (sl == 0u) o.chunk_lo[c] = c ? s : 0u;  // chunk 0 also owns the inactive ids in front of its first source
      bits |= 1u << i;
    } else if (k < PT_NCLS) {
      unsigned r = 0;
#pragma unroll
      for (int q = 1; q < PT_NCLS; q++)
        if (k == q) r = run[q]++;
      code = ((unsigned)k << PT_CLASS_SHIFT) | r;
      o.ids[k - 1][r] = s;
    }
    o.smap[s] = code;
  }
  if (base < a.n) o.src_bits8[base >> 3] = (uint8_t)bits;
}

// smap over the caller's (raw) column ids when the layout works in a relabelled vertex space (squished PageRank plan):
// colmap = exclusive scan of the live flags, so id c is live iff colmap[c + 1] > colmap[c]
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_smap_raw_kernel(const eoff_t *__restrict__ colmap, unsigned m_raw, const uint32_t *__restrict__ smap_l,
                   uint32_t *__restrict__ smap_raw) {
  const unsigned c = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (c >= m_raw) return;
  const eoff_t a = colmap[c], b = colmap[c + 1];
  smap_raw[c] = b > a ? smap_l[a] : PT_INACTIVE;
}

// rows: block counts of the rows that have entries
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_row_count_kernel(const eoff_t *__restrict__ rowptr, unsigned m, uint32_t *__restrict__ bc) {
  __shared__ unsigned s_n;
  if (threadIdx.x == 0) s_n = 0u;
  __syncthreads();
  const unsigned base = blockIdx.x * PT_VTILE + threadIdx.x * 8u;
  unsigned c = 0;
  if (base < m) {
    eoff_t prev = rowptr[base];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const unsigned r = base + (unsigned)i;
      if (r < m) {
        const eoff_t nx = rowptr[r + 1];
        c += nx > prev ? 1u : 0u;
        prev = nx;
      }
    }
  }
  const unsigned t = gdn_wave_sum(c);
  if (gdn_lane() == 0 && t) atomicAdd(&s_n, t);
  __syncthreads();
  if (threadIdx.x == 0) bc[blockIdx.x] = s_n;
}

struct PtRowOut {
  eoff_t *crp;         // n_dst + 1: first edge of the k-th row with entries
  eoff_t *bin_e;       // nbins + 1: first edge of every bin
  uint32_t *bin_lo;    // nbins + 1: first row id of every bin
  uint8_t *dst_bits8;
  uint32_t *rowstart;  // one bit per edge: set at the first edge of every row with entries
  unsigned per_b;
};
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_row_assign_kernel(const eoff_t *__restrict__ rowptr, unsigned m, const uint32_t *__restrict__ bc, PtRowOut o) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK];
  const unsigned base = blockIdx.x * PT_VTILE + threadIdx.x * 8u;
  eoff_t st[9];
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const unsigned r = base + (unsigned)i;
    st[i] = r <= m ? rowptr[r] : 0;
  }
  unsigned act = 0u;
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (base + (unsigned)i < m && st[i + 1] > st[i]) {
      act |= 1u << i;
      c++;
    }
  unsigned total;
  unsigned k = bc[blockIdx.x] + gdn_block_excl_scan(c, s_scan, &total);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (!((act >> i) & 1u)) continue;
    const unsigned r = base + (unsigned)i;
    o.crp[k] = st[i];
    const unsigned b = k / o.per_b;
    if (k - b * o.per_b == 0u) {
      o.bin_lo[b] = b ? r : 0u;
      o.bin_e[b] = st[i];
    }
    atomicOr(&o.rowstart[st[i] >> 5], 1u << (st[i] & 31u));
    k++;
  }
  if (base < m) o.dst_bits8[base >> 3] = (uint8_t)act;
}

static __global__ void pt_ends_kernel(uint32_t *chunk_lo, unsigned nchunks, unsigned n_src_ids, uint32_t *bin_lo, unsigned nbins,
                                      unsigned m_rows, eoff_t *bin_e, eoff_t *crp, eoff_t n_dst, eoff_t nnz, int have_src, int have_rows) {
  if (threadIdx.x == 0) {
    chunk_lo[nchunks] = n_src_ids;
    if (!have_src) chunk_lo[0] = 0u;
    bin_lo[nbins] = m_rows;
    bin_e[nbins] = nnz;
    crp[n_dst] = nnz;
    if (!have_rows) {
      bin_lo[0] = 0u;
      bin_e[0] = 0;
    }
  }
}

// exact marks of the columns that occur (only without trusted counts)
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_mark_kernel(const vid_t *__restrict__ colidx, eoff_t nnz, uint32_t *__restrict__ mark) {
  for (eoff_t e = (eoff_t)blockIdx.x * GDN_BLOCK + threadIdx.x; e < nnz; e += (eoff_t)gridDim.x * GDN_BLOCK)
    mark[__builtin_nontemporal_load(colidx + e)] = 1u;  // benign race: everybody stores 1
}
// exact counts as the input of the histogram kernels (negative = 0)
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_count16_kernel(const int32_t *__restrict__ deg, unsigned n, uint32_t *__restrict__ out) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < n) out[i] = deg[i] > 0 ? (uint32_t)deg[i] : 0u;
}

// ---- pt_keygen: the ONE gather pass ---------------------------------------------------------------------------------
struct PtKeyArgs {
  const vid_t *colidx;
  const uint32_t *smap;           // over the column ids of colidx
  const unsigned long long *rowstart;  // 64-bit view of the row-start bitmap
  const eoff_t *bin_e, *crp;
  eoff_t n_dst;
  unsigned per_b, nchunks;
  int log_chunk, ntiers;
  uint32_t *S;
  uint16_t *R;
  uint32_t *tile_cnt;   // nbins x nchunks
  uint32_t *tier_cnt;   // nbins x 8
  unsigned *errflag;
};
static __global__ void __launch_bounds__(PT_THREADS)
pt_keygen_kernel(PtKeyArgs a) {
  extern __shared__ unsigned s_tile[];
  __shared__ unsigned s_tier[8];
  const unsigned b = blockIdx.x;
  for (unsigned i = threadIdx.x; i < a.nchunks; i += PT_THREADS) s_tile[i] = 0u;
  if (threadIdx.x < 8) s_tier[threadIdx.x] = 0u;
  __syncthreads();
  const eoff_t E0 = a.bin_e[b], E1 = a.bin_e[b + 1];
  const unsigned w = threadIdx.x >> 6, lane = gdn_lane();
  unsigned bad = 0u;
  if (E1 > E0) {
    const eoff_t w0 = E0 >> 6, w1 = (E1 - 1) >> 6;  // 64-edge words of the bin, inclusive
    constexpr unsigned KW = PT_THREADS / 64;
    const eoff_t per = (w1 - w0 + KW) / KW;
    const eoff_t wa = w0 + (eoff_t)w * per, wb = wa + per < w1 + 1 ? wa + per : w1 + 1;
    if (wa < wb) {
      const eoff_t kb0 = (eoff_t)b * a.per_b, kb1 = kb0 + a.per_b < a.n_dst ? kb0 + a.per_b : a.n_dst;
      // rows of this bin that start in front of the wave's first edge
      eoff_t x = wa << 6;
      if (x < E0) x = E0;
      eoff_t lo = kb0, hi = kb1;
      while (lo < hi) {
        const eoff_t mid = lo + ((hi - lo) >> 1);
        if (a.crp[mid] < x) lo = mid + 1;
        else hi = mid;
      }
      eoff_t rb = lo;
      const unsigned long long le = lane == 63u ? ~0ull : ((2ull << lane) - 1ull);
      const unsigned long long lt = gdn_lanemask_lt();
      unsigned tcnt[PB_MAX_REC_TIERS];
#pragma unroll
      for (int t = 0; t < PB_MAX_REC_TIERS; t++) tcnt[t] = 0u;
      constexpr int UNR = 4;
      for (eoff_t wi = wa; wi < wb; wi += UNR) {
        unsigned long long bits[UNR];
        vid_t col[UNR];
        uint32_t code[UNR];
        bool ok[UNR];
#pragma unroll
        for (int r = 0; r < UNR; r++) {
          const eoff_t wj = wi + r, e = (wj << 6) + lane;
          ok[r] = wj < wb && e >= E0 && e < E1;
          bits[r] = wj < wb ? a.rowstart[wj] : 0ull;
          col[r] = ok[r] ? __builtin_nontemporal_load(a.colidx + e) : 0;
        }
#pragma unroll
        for (int r = 0; r < UNR; r++) code[r] = ok[r] ? a.smap[col[r]] : 0u;
#pragma unroll
        for (int r = 0; r < UNR; r++) {
          const eoff_t wj = wi + r;
          if (wj >= wb) break;  // wave-uniform
          unsigned long long bt = bits[r];
          if ((wj << 6) < E0) bt &= ~0ull << (E0 & 63u);
          if (((wj + 1) << 6) > E1) bt &= ~0ull >> (64u - (unsigned)(E1 & 63u));
          const unsigned rib = (unsigned)(rb + (eoff_t)__popcll(bt & le) - 1 - kb0);
          rb += (eoff_t)__popcll(bt);
          const eoff_t e = (wj << 6) + lane;
          const uint32_t c = code[r];
          const unsigned k = c >> PT_CLASS_SHIFT;
          if (ok[r]) {
            a.S[e] = c;
            a.R[e] = (uint16_t)rib;
            if (c == PT_INACTIVE) bad = 1u;
          }
          // tile counts: ONE LDS atomic per run of equal chunks in the wave (a hub row is hundreds of consecutive
          // edges into one chunk: 64 lanes on one counter serialise)
          const unsigned key = (ok[r] && k == 0u) ? (c >> a.log_chunk) : 0xFFFFFFFFu;
          const unsigned prev = (unsigned)__shfl_up((int)key, 1, 64);
          const bool head = lane == 0u || prev != key;
          const unsigned long long heads = __ballot(head);
          if (head && key != 0xFFFFFFFFu) {
            const unsigned long long after = lane == 63u ? 0ull : heads >> (lane + 1u);
            const unsigned len = after ? (unsigned)__ffsll((long long)after) : 64u - lane;
            atomicAdd(&s_tile[key], len);
          }
#pragma unroll
          for (int t = 0; t < PB_MAX_REC_TIERS; t++)
            if (t < a.ntiers) tcnt[t] += (unsigned)__popcll(__ballot(ok[r] && c != PT_INACTIVE && k == (unsigned)(t + 1)));
          (void)lt;
        }
      }
      if (lane == 0u) {
#pragma unroll
        for (int t = 0; t < PB_MAX_REC_TIERS; t++)
          if (t < a.ntiers && tcnt[t]) atomicAdd(&s_tier[t], tcnt[t]);
      }
    }
  }
  if (bad) *a.errflag = 1u;
  __syncthreads();
  for (unsigned i = threadIdx.x; i < a.nchunks; i += PT_THREADS) a.tile_cnt[(size_t)b * a.nchunks + i] = s_tile[i];
  if (threadIdx.x < 8) a.tier_cnt[(size_t)b * 8 + threadIdx.x] = s_tier[threadIdx.x];
}

// ---- offsets --------------------------------------------------------------------------------------------------------
struct PtTierPads {  // a tier's bin streams are padded to multiples of pad[t] records (a power of two)
  unsigned pad[PB_MAX_REC_TIERS];
};
// padded tile sizes in both tile orders + the (unpadded) sizes of the (bin, 8 chunks) segments of pt_split's output
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_tile_sizes_kernel(const uint32_t *__restrict__ tile_cnt, unsigned nchunks, unsigned nbins, unsigned pad, unsigned d1,
                     uint32_t *__restrict__ psz_c, uint32_t *__restrict__ psz_b, uint32_t *__restrict__ segsz) {
  const unsigned long long t = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;  // bin-major tile index
  if (t < (unsigned long long)nchunks * nbins) {
    const unsigned b = (unsigned)(t / nchunks), c = (unsigned)(t % nchunks);
    const uint32_t sz = (tile_cnt[t] + (pad - 1u)) & ~(pad - 1u);
    psz_b[t] = sz;
    psz_c[(unsigned long long)c * nbins + b] = sz;
  }
  if (t < (unsigned long long)d1 * nbins) {
    const unsigned b = (unsigned)(t / d1), d = (unsigned)(t % d1);
    uint32_t s = 0;
    for (unsigned k = 0; k < (1u << PT_LOW_BITS); k++) {
      const unsigned c = (d << PT_LOW_BITS) + k;
      if (c < nchunks) s += tile_cnt[(unsigned long long)b * nchunks + c];
    }
    segsz[t] = s;
  }
}
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_tier_sizes_kernel(const uint32_t *__restrict__ tier_cnt, unsigned nbins, int ntiers, uint32_t *__restrict__ tsz /*ntiers x nbins*/,
                     PtTierPads pads) {
  const unsigned b = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (b >= nbins) return;
  for (int t = 0; t < ntiers; t++) tsz[(size_t)t * nbins + b] = (tier_cnt[(size_t)b * 8 + t] + (pads.pad[t] - 1u)) & ~(pads.pad[t] - 1u);
}
// Interleaved record streams (PbTierSet::Tier::interleaved): inside every block of 256 records (streams are padded to whole
// blocks) record r of the sorted order sits at (r % 64) * 4 + r / 64, so that ONE 16-byte load per lane hands lane l the
// records l, 64 + l, 128 + l, 192 + l -- phase B then issues a quarter of the record loads, and its j-th table read still
// covers 64 CONSECUTIVE records of the source-sorted stream (four consecutive records per lane spread the 64 reads of an
// instruction over four times as many table lines).  In place, a wave per block: all four loads of every lane have
// returned before the wave's store issues.
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_interleave_kernel(uint32_t *__restrict__ rec, unsigned long long nblk) {
  typedef unsigned pt_u32x4 __attribute__((ext_vector_type(4)));
  const unsigned lane = gdn_lane();
  for (unsigned long long q = (unsigned long long)blockIdx.x * GDN_WAVES_PER_BLOCK + (threadIdx.x >> 6); q < nblk;
       q += (unsigned long long)gridDim.x * GDN_WAVES_PER_BLOCK) {
    uint32_t *base = rec + q * 256ull;
    pt_u32x4 v;
    v.x = base[lane];
    v.y = base[64u + lane];
    v.z = base[128u + lane];
    v.w = base[192u + lane];
    __builtin_amdgcn_s_waitcnt(0);  // (the data dependence already makes the compiler wait; kept explicit)
    __builtin_amdgcn_wave_barrier();
    reinterpret_cast<pt_u32x4 *>(base)[lane] = v;
  }
}

// ---- stable partition of a range of items by a digit, staged through LDS ---------------------------------------------
// lanes of the wave whose (valid) digit equals mine
__device__ __forceinline__ unsigned long long pt_match(unsigned d, bool valid, int bits) {
  unsigned long long peers = __ballot(valid);
  for (int b = 0; b < bits; b++) {
    const bool one = (d >> b) & 1u;
    const unsigned long long m = __ballot(one && valid);
    peers &= one ? m : ~m;
  }
  return peers;
}
// exclusive scan over the PT_PTHREADS threads of a partition workgroup; scratch = PT_WAVES + 1 unsigned
__device__ __forceinline__ unsigned pt_block_excl_scan(unsigned v, unsigned *scratch, unsigned *total) {
  const unsigned incl = gdn_wave_incl_scan(v);
  const unsigned w = threadIdx.x >> 6;
  __syncthreads();
  if (gdn_lane() == 63) scratch[w] = incl;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < PT_WAVES; i++) {
    const unsigned x = scratch[i];
    if ((unsigned)i < w) base += x;
    tot += x;
  }
  *total = tot;
  return base + incl - v;
}
struct PtStage {
  uint32_t *stage;           // PT_STEP items in digit order
  uint32_t *vstage;          // (value-carrying partitions: a second 32-bit word per item, e.g. SpMV's Ax) or nullptr
  unsigned short *dig;       // their digits
  unsigned *wc;              // PT_WAVES x nd: per wave and digit, count -> exclusive prefix over the waves
  unsigned *start, *tot;     // nd: first position of a digit in the staged order, its count in this step
  unsigned long long *goff;  // nd: where the digit's next item goes in `out`
  unsigned *scr;             // PT_WAVES + 1 (pt_block_excl_scan)
};
__device__ __forceinline__ PtStage pt_stage_carve(unsigned char *lds, unsigned nd, bool val = false) {
  PtStage s;
  s.stage = reinterpret_cast<uint32_t *>(lds);
  s.vstage = nullptr;
  if (val) {
    s.vstage = reinterpret_cast<uint32_t *>(lds + (size_t)PT_STEP * 4);
    lds += (size_t)PT_STEP * 4;
  }
  s.goff = reinterpret_cast<unsigned long long *>(lds + (size_t)PT_STEP * 4);
  s.start = reinterpret_cast<unsigned *>(s.goff + nd);
  s.tot = s.start + nd;
  s.scr = s.tot + nd;
  s.wc = s.scr + PT_WAVES + 2;
  s.dig = reinterpret_cast<unsigned short *>(s.wc + (size_t)PT_WAVES * nd);
  return s;
}
static inline size_t pt_stage_bytes(unsigned nd, bool val = false) {
  return (size_t)PT_STEP * (val ? 8 : 4) + (size_t)nd * 16 + (PT_WAVES + 2) * 4 + (size_t)PT_STEP * 2 + (size_t)PT_WAVES * nd * 4 + 64;
}
// Walks items [0, n) in steps of PT_STEP.  load(i, item, digit) for i < n.  The caller has set st.goff[d] (behind a
// barrier) and zeroed st.wc.  Every thread of the workgroup must call.
// Rank of an item among the items of its digit: a wave owns 1024 consecutive items of a step and counts them in ITS row of
// st.wc.  STABLE = false (default): one LDS atomic with return per item -- sixteen independent ones in flight per lane;
// items of one wave instruction that share a digit take their slots in the order the LDS unit serialises the conflict
// (lane order on this hardware: the partition came out stable in every comparison with the other form, but nothing
// documents it, and no result depends on the order inside a tile or a record stream).  STABLE = true (GDN_PT_STABLE=1):
// ballot matching, the leader lane adds the group's count -- stable by construction, ~10 VALU instructions per digit bit
// and item, and one LDS round trip per item in sequence (RMAT-27: pt_split 23.8 ms, pt_radix 31 ms).
// VAL: every item carries a second 32-bit word (load(i, item, digit, value)), written to vout at the item's position.
template <bool STABLE, bool VAL, class Load>
__device__ __forceinline__ void pt_partition(unsigned long long n, unsigned nd, int dbits, uint32_t *__restrict__ out,
                                             uint32_t *__restrict__ vout, const PtStage &st, Load load) {
  const unsigned w = threadIdx.x >> 6, lane = gdn_lane();
  const unsigned long long lt = gdn_lanemask_lt();
  unsigned *wcw = st.wc + (size_t)w * nd;
  for (unsigned long long base = 0; base < n; base += PT_STEP) {
    const unsigned cnt = (unsigned)(n - base < (unsigned long long)PT_STEP ? n - base : (unsigned long long)PT_STEP);
    uint32_t it[PT_IPT], vl[VAL ? PT_IPT : 1];
    unsigned short dg[PT_IPT], rk[PT_IPT];
#pragma unroll
    for (int j = 0; j < PT_IPT; j++) {
      const unsigned i = w * (PT_IPT * 64u) + (unsigned)j * 64u + lane;  // a wave owns 1024 consecutive items of the step
      it[j] = 0u;
      unsigned d = 0u;
      uint32_t v = 0u;
      if (i < cnt) load(base + i, it[j], d, v);
      if constexpr (VAL) vl[j] = v;
      dg[j] = (unsigned short)d;
    }
    if constexpr (STABLE) {
#pragma unroll
      for (int j = 0; j < PT_IPT; j++) {
        const unsigned i = w * (PT_IPT * 64u) + (unsigned)j * 64u + lane;
        const bool valid = i < cnt;
        const unsigned d = dg[j];
        const unsigned long long peers = pt_match(d, valid, dbits);
        const unsigned r = (unsigned)__popcll(peers & lt);
        unsigned old = 0u;
        if (valid && r == 0u) {
          old = wcw[d];
          wcw[d] = old + (unsigned)__popcll(peers);
        }
        const int leader = valid ? __ffsll((long long)peers) - 1 : (int)lane;
        old = (unsigned)__shfl((int)old, leader, 64);
        rk[j] = (unsigned short)(old + r);
      }
    } else {
#pragma unroll
      for (int j = 0; j < PT_IPT; j++) {
        const unsigned i = w * (PT_IPT * 64u) + (unsigned)j * 64u + lane;
        rk[j] = i < cnt ? (unsigned short)atomicAdd(&wcw[dg[j]], 1u) : (unsigned short)0;
      }
    }
    __syncthreads();
    // per digit (two per thread): counts of the waves -> exclusive prefix over the waves, total of the step
    unsigned mine[2] = {0u, 0u};
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const unsigned d = 2u * threadIdx.x + (unsigned)q;
      if (d < nd) {
        unsigned acc = 0u;
#pragma unroll
        for (int ww = 0; ww < PT_WAVES; ww++) {
          const unsigned c = st.wc[(size_t)ww * nd + d];
          st.wc[(size_t)ww * nd + d] = acc;
          acc += c;
        }
        mine[q] = acc;
        st.tot[d] = acc;
      }
    }
    unsigned total;
    const unsigned ex = pt_block_excl_scan(mine[0] + mine[1], st.scr, &total);
    if (2u * threadIdx.x < nd) st.start[2u * threadIdx.x] = ex;
    if (2u * threadIdx.x + 1u < nd) st.start[2u * threadIdx.x + 1u] = ex + mine[0];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PT_IPT; j++) {
      const unsigned i = w * (PT_IPT * 64u) + (unsigned)j * 64u + lane;
      if (i < cnt) {
        const unsigned d = dg[j];
        const unsigned pos = st.start[d] + wcw[d] + rk[j];
        st.stage[pos] = it[j];
        if constexpr (VAL) st.vstage[pos] = vl[j];
        st.dig[pos] = (unsigned short)d;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PT_IPT; j++) {
      const unsigned pos = (unsigned)j * PT_PTHREADS + threadIdx.x;
      if (pos < cnt) {
        const unsigned d = st.dig[pos];
        out[st.goff[d] + (pos - st.start[d])] = st.stage[pos];
        if constexpr (VAL) vout[st.goff[d] + (pos - st.start[d])] = st.vstage[pos];
      }
    }
    __syncthreads();
    for (unsigned d = threadIdx.x; d < nd; d += PT_PTHREADS) {
      st.goff[d] += st.tot[d];
#pragma unroll
      for (int ww = 0; ww < PT_WAVES; ww++) st.wc[(size_t)ww * nd + d] = 0;
    }
    __syncthreads();
  }
}

// pt_split: a bin's edges -> main-layout items by (chunk >> 3) and tier records by tier, all into X
struct PtSplitArgs {
  const uint32_t *S;
  const uint16_t *R;
  const eoff_t *bin_e;
  const eoff_t *segoff;                   // nbins x d1 (+1): item offsets of the (bin, digit) segments in X
  const eoff_t *tier_ptr[PB_MAX_REC_TIERS];  // nbins + 1 record offsets (multiples of 16) per tier
  eoff_t tier_base[PB_MAX_REC_TIERS];     // where tier t's records start in X
  const uint32_t *order;                  // bins, largest first
  uint32_t *X;
  const uint32_t *EV;                     // VAL: one 32-bit word per edge in CSR order (SpMV's Ax) ...
  uint32_t *XV;                           // ... moved like the items, into XV
  unsigned d1;
  int ntiers, log_chunk, dbits;
};
template <bool STABLE, bool VAL>
static __global__ void __launch_bounds__(PT_PTHREADS, 4)
pt_split_kernel(PtSplitArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const unsigned nd = a.d1 + (unsigned)a.ntiers;
  const PtStage st = pt_stage_carve(s_raw, nd, VAL);
  const unsigned b = a.order[blockIdx.x];
  for (unsigned d = threadIdx.x; d < nd; d += PT_PTHREADS) {
    st.goff[d] = d < a.d1 ? a.segoff[(size_t)b * a.d1 + d] : a.tier_base[d - a.d1] + a.tier_ptr[d - a.d1][b];
#pragma unroll
    for (int ww = 0; ww < PT_WAVES; ww++) st.wc[(size_t)ww * nd + d] = 0;
  }
  __syncthreads();
  const eoff_t E0 = a.bin_e[b], E1 = a.bin_e[b + 1];
  const uint32_t *S = a.S + E0;
  const uint16_t *R = a.R + E0;
  const unsigned slot_mask = (1u << a.log_chunk) - 1u;
  const int lc = a.log_chunk;
  const unsigned d1 = a.d1;
  const uint32_t *EV = VAL ? a.EV + E0 : nullptr;
  pt_partition<STABLE, VAL>(E1 - E0, nd, a.dbits, a.X, a.XV, st, [&](unsigned long long i, uint32_t &item, unsigned &d, uint32_t &val) {
    const uint32_t c = __builtin_nontemporal_load(S + i);
    const unsigned row = __builtin_nontemporal_load(R + i);
    if constexpr (VAL) val = __builtin_nontemporal_load(EV + i);
    const unsigned k = c >> PT_CLASS_SHIFT;
    if (k == 0u) {
      const unsigned chunk = c >> lc;
      d = chunk >> PT_LOW_BITS;
      item = ((c & slot_mask) << 17) | (row << PT_LOW_BITS) | (chunk & ((1u << PT_LOW_BITS) - 1u));
    } else {
      d = d1 + k - 1u;
      item = ((c & PT_IDX_MASK) << PB_MID_ROW_BITS) | row;
    }
  });
}

// pt_tiles: a wave per (bin, 8 chunks) segment of X -> V (bin-major) and U (chunk-major)
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_tiles_kernel(const uint32_t *__restrict__ X, const eoff_t *__restrict__ segoff, unsigned d1, unsigned nbins,
                unsigned nchunks, const eoff_t *__restrict__ pu, const eoff_t *__restrict__ pv, uint16_t *__restrict__ U,
                uint16_t *__restrict__ V,
                // nullable: the items' value words (pt_split_kernel<., true>) and where they go -- next to U, chunk-major
                const uint32_t *__restrict__ XV = nullptr, uint32_t *__restrict__ AU = nullptr) {
  const unsigned long long seg = ((unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x) >> 6;
  if (seg >= (unsigned long long)d1 * nbins) return;
  const unsigned b = (unsigned)(seg / d1), d = (unsigned)(seg % d1), lane = gdn_lane();
  const eoff_t s0 = segoff[seg], s1 = segoff[seg + 1];
  if (s1 == s0) return;
  // lane k < 8 holds the tile offsets of chunk 8 d + k
  eoff_t offu = 0, offv = 0;
  const unsigned cl = (d << PT_LOW_BITS) + (lane & 7u);
  if (cl < nchunks) {
    offu = pu[(unsigned long long)cl * nbins + b];
    offv = pv[(unsigned long long)b * nchunks + cl];
  }
  const unsigned long long lt = gdn_lanemask_lt();
  unsigned basek[1 << PT_LOW_BITS];
#pragma unroll
  for (int k = 0; k < (1 << PT_LOW_BITS); k++) basek[k] = 0u;
  for (eoff_t i0 = s0; i0 < s1; i0 += 64) {
    const eoff_t i = i0 + lane;
    const bool valid = i < s1;
    const uint32_t item = valid ? __builtin_nontemporal_load(X + i) : 0u;
    const uint32_t val = (valid && XV) ? __builtin_nontemporal_load(XV + i) : 0u;
    const unsigned low = item & ((1u << PT_LOW_BITS) - 1u);
    unsigned rank = 0u;
#pragma unroll
    for (int k = 0; k < (1 << PT_LOW_BITS); k++) {
      const unsigned long long mk = __ballot(valid && low == (unsigned)k);
      if (low == (unsigned)k) rank = basek[k] + (unsigned)__popcll(mk & lt);
      basek[k] += (unsigned)__popcll(mk);
    }
    const eoff_t ou = (eoff_t)__shfl((long long)offu, (int)low, 64), ov = (eoff_t)__shfl((long long)offv, (int)low, 64);
    if (valid) {
      U[ou + rank] = (uint16_t)(item >> 17);
      V[ov + rank] = (uint16_t)((item >> PT_LOW_BITS) & ((1u << PB_MID_ROW_BITS) - 1u));
      if (AU) AU[ou + rank] = val;
    }
  }
}

// pt_radix: one stable pass over the records of every (bin) segment of ONE tier by bits [shift, shift + bits) of the record
struct PtRadixArgs {
  const uint32_t *in;
  uint32_t *out;
  const eoff_t *ptr;          // nbins + 1 record offsets (the same in `in` and `out`)
  const uint32_t *tier_cnt;   // nbins x 8
  const uint32_t *order;
  int tier, shift, bits;
  int last;                   // final pass: pad records behind the segment
  uint32_t zrec;
  const uint32_t *vin;        // VAL: the records' value words (same offsets as the records)
  uint32_t *vout;
};
template <bool STABLE, bool VAL>
static __global__ void __launch_bounds__(PT_PTHREADS, 4)
pt_radix_kernel(PtRadixArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const unsigned nd = 1u << a.bits;
  const PtStage st = pt_stage_carve(s_raw, nd, VAL);
  const unsigned b = a.order[blockIdx.x];
  const eoff_t p0 = a.ptr[b];
  const unsigned n = a.tier_cnt[(size_t)b * 8 + a.tier];
  const uint32_t *in = a.in + p0;
  const unsigned mask = nd - 1u;
  // histogram of the segment (st.tot doubles as the counter array; eight loads in flight per thread)
  for (unsigned d = threadIdx.x; d < nd; d += PT_PTHREADS) {
    st.tot[d] = 0u;
#pragma unroll
    for (int ww = 0; ww < PT_WAVES; ww++) st.wc[(size_t)ww * nd + d] = 0;
  }
  __syncthreads();
  for (unsigned i0 = threadIdx.x; i0 < n; i0 += 8u * PT_PTHREADS) {
    uint32_t v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const unsigned i = i0 + (unsigned)r * PT_PTHREADS;
      v[r] = i < n ? in[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < 8; r++)
      if (i0 + (unsigned)r * PT_PTHREADS < n) atomicAdd(&st.tot[(v[r] >> a.shift) & mask], 1u);
  }
  __syncthreads();
  {
    const unsigned d0 = 2u * threadIdx.x, d1 = d0 + 1u;
    const unsigned m0 = d0 < nd ? st.tot[d0] : 0u, m1 = d1 < nd ? st.tot[d1] : 0u;
    unsigned total;
    const unsigned ex = pt_block_excl_scan(m0 + m1, st.scr, &total);
    if (d0 < nd) st.goff[d0] = p0 + ex;
    if (d1 < nd) st.goff[d1] = p0 + ex + m0;
  }
  __syncthreads();
  const int shift = a.shift;
  const uint32_t *vin = VAL ? a.vin + p0 : nullptr;
  pt_partition<STABLE, VAL>(n, nd, a.bits, a.out, a.vout, st, [&](unsigned long long i, uint32_t &item, unsigned &d, uint32_t &val) {
    item = in[i];
    if constexpr (VAL) val = vin[i];
    d = (item >> shift) & mask;
  });
  if (a.last) {
    const unsigned npad = (unsigned)(a.ptr[b + 1] - p0);  // the stream's padded length (pt_tier_sizes_kernel)
    for (unsigned i = n + threadIdx.x; i < npad; i += PT_PTHREADS) {
      a.out[p0 + i] = a.zrec;
      if constexpr (VAL) a.vout[p0 + i] = 0u;  // the pad records' factor: 0.0f
    }
  }
}

// ---- small kernels of the offsets phase (the forms of gdn_build.hip's pb_build on counts instead of sorted keys)
static __global__ void __launch_bounds__(GDN_BLOCK)
pt_sizes_from_ptr_kernel(const eoff_t *__restrict__ ptr, unsigned n, eoff_t *__restrict__ sz) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < n) sz[i] = ptr[i + 1] - ptr[i];
}

#define PT_CHECK_PTR(p)                                              \
  do {                                                               \
    if (!(p)) {                                                      \
      gdn_set_error("layout build: scratch arena too small (%s:%d)", __FILE__, __LINE__); \
      return GDN_ERR_INVALID;                                        \
    }                                                                \
  } while (0)

// GDN_ERR_UNSUPPORTED-like: returns 1 (positive) when the shape is outside the limits above and nothing was built
static int pb_build_tiered(const PbTieredArgs &a, PbPlan &p, PbTierSet &ts) {
  const auto t_begin = std::chrono::steady_clock::now();
  const bool trace = gdn_xoption("GDN_PB_TRACE") != nullptr;
  const bool stable = gdn_xoption("GDN_PT_STABLE") != nullptr;  // ranks by ballot matching (see pt_partition)
  auto t_last = t_begin;
  auto phase = [&](const char *name) {  // GDN_PB_TRACE: wall time of every phase (synchronises: the timings perturb)
    if (!trace) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[pb_build_tiered]   %-22s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  GDN_REQUIRE(a.rowptr && (a.colidx || a.nnz == 0) && a.m_rows >= 0 && a.m_global > 0, "layout build: arguments");
  GDN_REQUIRE(!a.colmap || a.src_count, "layout build: a column map needs source counts");
  const int lc = a.log_chunk, lb = a.log_bin;
  GDN_REQUIRE(lc >= 8 && lc <= 15 && lb >= 8 && lb <= PB_MID_ROW_BITS, "layout build: log_chunk / log_bin");
  GDN_REQUIRE(a.pad >= (1u << a.log_group) && a.pad <= 128 && (a.pad & (a.pad - 1)) == 0 && a.log_group >= 3, "pad / log_group");
  const unsigned mL = (unsigned)a.m_global, mR = (unsigned)a.m_rows;
  const unsigned m_raw = a.colmap ? (unsigned)a.m_raw : mL;
  const unsigned long long n = a.nnz;
  if ((uint64_t)((mL + (1u << lc) - 1) >> lc) > PT_MAX_CHUNKS) return 1;
  const unsigned grp = 1u << a.log_group;
  const unsigned nbS = (mL + PT_VTILE - 1) / PT_VTILE, nbR = (mR + PT_VTILE - 1) / PT_VTILE;
  ts.n = 0;
  // ---- arena 1: vertex phase + the per-edge codes
  PtArena A1;
  {
    size_t bytes = 0;
    bytes += pt_pad256((size_t)mL * 4) * 3;                  // cnt16, mark, smap_l
    bytes += a.colmap ? pt_pad256((size_t)m_raw * 4) : 0;    // smap_raw
    bytes += pt_pad256((size_t)PT_NCLS * nbS * 4) + pt_pad256((size_t)nbR * 4) + 4096;
    bytes += pt_pad256(((size_t)mR + 2) * 8);                // crp
    bytes += pt_pad256(((size_t)(n >> 5) + 4) * 4);          // rowstart
    bytes += pt_pad256((size_t)n * 4) + pt_pad256((size_t)n * 2);  // S, R
    bytes += pt_pad256((PB_HUB_BUCKETS + PB_LIN_BINS + 64) * 4);
    GDN_TRY(A1.init(bytes));
  }
  uint32_t *cnt16 = A1.get<uint32_t>(mL), *mark = A1.get<uint32_t>(mL), *smap_l = A1.get<uint32_t>(mL);
  uint32_t *smap_raw = a.colmap ? A1.get<uint32_t>(m_raw) : smap_l;
  uint32_t *bcS = A1.get<uint32_t>((size_t)PT_NCLS * nbS), *bcR = A1.get<uint32_t>(nbR);
  uint32_t *totals = A1.get<uint32_t>(16);
  unsigned *hist = A1.get<unsigned>(PB_HUB_BUCKETS + PB_LIN_BINS);
  unsigned *errflag = A1.get<unsigned>(4);
  eoff_t *crp = A1.get<eoff_t>((size_t)mR + 2);
  uint32_t *rowstart = A1.get<uint32_t>((size_t)(n >> 5) + 4);
  uint32_t *S = A1.get<uint32_t>(n);
  uint16_t *R = A1.get<uint16_t>(n);
  PT_CHECK_PTR(R);
  PT_CHECK_PTR(S);
  phase("arena 1");
  GDN_TRY(pt_zero(hist, (PB_HUB_BUCKETS + PB_LIN_BINS) * 4));
  GDN_TRY(pt_zero(errflag, 16));
  GDN_TRY(pt_zero(rowstart, ((size_t)(n >> 5) + 4) * 4));
  // ---- source counts (sampled units) and their histograms; active rows
  const bool want_tiers = a.tiers && lb <= PB_MID_ROW_BITS;
  if (a.src_count) {
    if (want_tiers) hipLaunchKernelGGL(pt_count16_kernel, dim3(gdn_nblocks(mL)), dim3(GDN_BLOCK), 0, 0, a.src_count, mL, cnt16);
  } else {
    GDN_TRY(pt_zero(mark, (size_t)mL * 4));
    GDN_TRY(pt_zero(cnt16, (size_t)mL * 4));
    if (n) {
      const unsigned long long gb = (n + GDN_BLOCK * 8ull - 1) / (GDN_BLOCK * 8ull);
      hipLaunchKernelGGL(pt_mark_kernel, dim3((unsigned)(gb > 262144ull ? 262144ull : gb)), dim3(GDN_BLOCK), 0, 0, a.colidx, (eoff_t)n, mark);
      if (want_tiers) {
        const uint64_t sampled = ((uint64_t)mR + (1u << PB_HUB_SAMPLE_LOG) - 1) >> PB_HUB_SAMPLE_LOG;
        hipLaunchKernelGGL(pb_hub_sample_kernel, dim3(gdn_nblocks(sampled * 64)), dim3(GDN_BLOCK), 0, 0, a.rowptr, a.colidx, (int32_t)mR, cnt16);
      }
    }
  }
  if (want_tiers) {
    hipLaunchKernelGGL(pb_hub_hist_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, cnt16, (size_t)mL, hist);
    if (a.max_mid > 0) hipLaunchKernelGGL(pb_hub_hist_lin_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, cnt16, (size_t)mL, hist + PB_HUB_BUCKETS);
  }
  hipLaunchKernelGGL(pt_row_count_kernel, dim3(nbR ? nbR : 1), dim3(GDN_BLOCK), 0, 0, a.rowptr, mR, bcR);
  hipLaunchKernelGGL(pt_scan_rows_kernel, dim3(1), dim3(GDN_BLOCK), 0, 0, bcR, nbR, totals + 8);
  GDN_HIP(hipGetLastError());
  std::vector<unsigned> h_hist(PB_HUB_BUCKETS + PB_LIN_BINS);
  unsigned h_tot[16];
  GDN_HIP(hipMemcpy(h_hist.data(), hist, h_hist.size() * 4, hipMemcpyDeviceToHost));
  GDN_HIP(hipMemcpy(h_tot + 8, totals + 8, 4, hipMemcpyDeviceToHost));
  const uint64_t n_dst = h_tot[8];
  phase("counts + histograms");
  PtSrcArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.src_count = a.src_count;
  sa.cnt16 = cnt16;
  sa.mark = mark;
  sa.n = mL;
  sa.ntiers = 0;
  bool has_hub = false;
  if (want_tiers) {
    unsigned thr[1 + PB_MAX_MID];
    int nt = 0;
    const uint64_t nbins_est = ((n_dst + (1ull << lb) - 1) >> lb) + 1;
    // exact counts cut the tiers at single degrees (the linear histogram covers degrees < 4095: every mid-tier threshold
    // of a graph with <= 2^17 bins); sampled ones stand for 16 edges each
    pb_choose_tiers(h_hist.data(), a.max_mid > 0 ? h_hist.data() + PB_HUB_BUCKETS : nullptr, nbins_est, n, a.max_mid, a.min16, thr, &nt,
                    a.src_count ? 0 : PB_HUB_SAMPLE_LOG);
    has_hub = thr[0] != 0xFFFFFFFFu;
    // classes in use, in descending threshold order (a missing hub class is dropped: the tiers are numbered 1.. here)
    for (int t = has_hub ? 0 : 1; t < nt; t++) sa.thr[sa.ntiers++] = thr[t];
  }
  hipLaunchKernelGGL(pt_src_count_kernel, dim3(nbS), dim3(GDN_BLOCK), 0, 0, sa, nbS, bcS);
  hipLaunchKernelGGL(pt_scan_rows_kernel, dim3(PT_NCLS), dim3(GDN_BLOCK), 0, 0, bcS, nbS, totals);
  GDN_HIP(hipGetLastError());
  GDN_HIP(hipMemcpy(h_tot, totals, PT_NCLS * 4, hipMemcpyDeviceToHost));
  const uint64_t n_src0 = h_tot[0];
  phase("class counts");
  if (trace) {
    fprintf(stderr, "[pb_build_tiered]   rows with entries %llu, sources per class:", (unsigned long long)n_dst);
    for (int k = 0; k < PT_NCLS; k++) fprintf(stderr, " %u", h_tot[k]);
    fprintf(stderr, "; thresholds:");
    for (int t = 0; t < sa.ntiers; t++) fprintf(stderr, " %u", sa.thr[t]);
    fprintf(stderr, " (%s counts)\n", a.src_count ? "exact" : "sampled");
  }
  // tiers that came out empty are dropped from the back (the histogram counted them, so only when nothing qualifies)
  while (sa.ntiers > 0 && h_tot[sa.ntiers] == 0) sa.ntiers--;
  for (int t = 0; t < sa.ntiers; t++) {
    const unsigned cap = (t == 0 && has_hub) ? (1u << PB_HUB_LOG) : PB_MID_MAX;
    if (h_tot[1 + t] > cap || h_tot[1 + t] == 0) {  // cannot happen: the histograms counted them
      gdn_set_error("layout build: tier %d holds %u sources (cap %u)", t, h_tot[1 + t], cap);
      return GDN_ERR_INVALID;
    }
  }
  const int ntiers = sa.ntiers;
  // ---- slices (whole rounds of workgroups, see pb_build)
  // (GDN_PB_BALANCE_ALL, experiments build: slices of ANY size spread over whole rounds of workgroups -- mid-size graphs, whose
  // chunks are 2^14 sources and whose 366 workgroups are 1.43 rounds, tools/pr_midsize.py)
  // Round 6: source CHUNKS of any size are spread over whole rounds (a chunk's workgroup fills a CU whatever its slice: 1024
  // threads) -- the LJ-like stand-in's 366 chunks of 2^14 sources were 1.43 rounds of phase A in the time of two: 512 chunks,
  // phase A 0.074 -> 0.058 ms, the iteration 0.177 -> 0.160 ms = 0.50 -> 0.55 of the roofline, gdn_pr 5.28 -> 4.87 ms
  // (profiles/r06_pr_midsize.txt).  Bins keep their rule (the same spreading measured no gain for shards' bins).
  const bool bal_all = gdn_xoption("GDN_PB_BALANCE_ALL") != nullptr;  // (experiments build: the bins too)
  const char *bce = gdn_xoption("GDN_PB_BALANCE_CHUNKS");
  const bool bal_chunks = !(bce && bce[0] == '0');
  const uint64_t per_c = pb_slots_per_slice(n_src0, lc, (bal_all || bal_chunks) ? lc : PB_MAX_LOG_CHUNK),
                 per_b = pb_slots_per_slice(n_dst, lb, bal_all ? lb : a.bin_balance_log);
  unsigned nchunks = (unsigned)((n_src0 + per_c - 1) / per_c), nbins = (unsigned)((n_dst + per_b - 1) / per_b);
  if (nchunks == 0) nchunks = 1;
  if (nbins == 0) nbins = 1;
  if (nchunks > PT_MAX_CHUNKS) return 1;
  const unsigned d1 = (nchunks + (1u << PT_LOW_BITS) - 1) >> PT_LOW_BITS;
  const unsigned nd_split = d1 + (unsigned)ntiers;
  if (nd_split > PT_MAX_DIGITS) return 1;
  p.m_local = a.m_rows;
  p.m_global = a.m_global;
  p.log_chunk = lc;
  p.log_bin = lb;
  p.compact = true;
  p.chunk_slots = (unsigned)per_c;
  p.log_group = a.log_group;
  p.nchunks = nchunks;
  p.nbins = nbins;
  const unsigned long long ntiles = (unsigned long long)nchunks * nbins;
  GDN_TRY(p.src_bits.alloc(((size_t)mL + 31) / 32 + 1));
  GDN_TRY(p.dst_bits.alloc(((size_t)mR + 31) / 32 + 1));
  GDN_TRY(p.chunk_lo.alloc((size_t)nchunks + 1));
  GDN_TRY(p.bin_lo.alloc((size_t)nbins + 1));
  GDN_TRY(p.chunk_ptr.alloc((size_t)nchunks + 1));
  GDN_TRY(p.bin_ptr.alloc((size_t)nbins + 1));
  GDN_TRY(p.errflag.alloc(1));
  GDN_HIP(hipMemsetAsync(p.errflag.p, 0, sizeof(unsigned), 0));
  GDN_HIP(hipMemsetAsync(p.src_bits.p, 0, (((size_t)mL + 31) / 32 + 1) * 4, 0));
  GDN_HIP(hipMemsetAsync(p.dst_bits.p, 0, (((size_t)mR + 31) / 32 + 1) * 4, 0));
  GDN_HIP(hipMemsetAsync(p.chunk_lo.p, 0, ((size_t)nchunks + 1) * 4, 0));
  GDN_HIP(hipMemsetAsync(p.bin_lo.p, 0, ((size_t)nbins + 1) * 4, 0));
  for (int t = 0; t < ntiers; t++) {
    ts.t[t].n_src = h_tot[1 + t];
    GDN_TRY(ts.t[t].ids.alloc(h_tot[1 + t]));
    GDN_TRY(ts.t[t].bin_ptr.alloc((size_t)nbins + 1));
  }
  // ---- arena 2: tile tables
  PtArena A2;
  {
    size_t bytes = 0;
    bytes += pt_pad256(ntiles * 4) * 3;                                  // tile_cnt, psz_c, psz_b
    bytes += pt_pad256((ntiles + 1) * 8) * 2;                            // pu, pv
    bytes += pt_pad256((size_t)d1 * nbins * 4) + pt_pad256(((size_t)d1 * nbins + 1) * 8);  // segsz, segoff
    bytes += pt_pad256((size_t)nbins * 8 * 4) + pt_pad256((size_t)PB_MAX_REC_TIERS * nbins * 4);
    bytes += pt_pad256(((size_t)nbins + 2) * 8) * 3 + pt_pad256((size_t)nbins * 4) + pt_pad256(((size_t)nchunks + 1) * 8);
    bytes += pt_pad256((ntiles / 2048 + 16) * 8) * 2;
    GDN_TRY(A2.init(bytes, 2));
  }
  uint32_t *tile_cnt = A2.get<uint32_t>(ntiles), *psz_c = A2.get<uint32_t>(ntiles), *psz_b = A2.get<uint32_t>(ntiles);
  eoff_t *pu = A2.get<eoff_t>(ntiles + 1), *pv = A2.get<eoff_t>(ntiles + 1);
  uint32_t *segsz = A2.get<uint32_t>((size_t)d1 * nbins);
  eoff_t *segoff = A2.get<eoff_t>((size_t)d1 * nbins + 1);
  uint32_t *tier_cnt = A2.get<uint32_t>((size_t)nbins * 8), *tsz = A2.get<uint32_t>((size_t)PB_MAX_REC_TIERS * nbins);
  eoff_t *bin_e = A2.get<eoff_t>((size_t)nbins + 2), *d_du = A2.get<eoff_t>((size_t)nchunks + 1), *d_dv = A2.get<eoff_t>((size_t)nbins + 2);
  eoff_t *bin_sz = A2.get<eoff_t>((size_t)nbins + 2);
  uint32_t *order = A2.get<uint32_t>(nbins);
  eoff_t *ws = A2.get<eoff_t>(ntiles / 2048 + 16);
  PT_CHECK_PTR(ws);
  GDN_TRY(pt_zero(bin_e, ((size_t)nbins + 2) * 8));
  phase("plan arrays + arena 2");
  // ---- ranks
  {
    PtSrcOut so;
    memset(&so, 0, sizeof(so));
    so.smap = smap_l;
    for (int t = 0; t < ntiers; t++) so.ids[t] = ts.t[t].ids.p;
    so.src_bits8 = reinterpret_cast<uint8_t *>(p.src_bits.p);
    so.chunk_lo = p.chunk_lo.p;
    so.per_c = (unsigned)per_c;
    so.log_chunk = lc;
    hipLaunchKernelGGL(pt_src_assign_kernel, dim3(nbS), dim3(GDN_BLOCK), 0, 0, sa, nbS, bcS, so);
    if (a.colmap) hipLaunchKernelGGL(pt_smap_raw_kernel, dim3(gdn_nblocks(m_raw)), dim3(GDN_BLOCK), 0, 0, a.colmap, m_raw, smap_l, smap_raw);
    PtRowOut ro;
    ro.crp = crp;
    ro.bin_e = bin_e;
    ro.bin_lo = p.bin_lo.p;
    ro.dst_bits8 = reinterpret_cast<uint8_t *>(p.dst_bits.p);
    ro.rowstart = rowstart;
    ro.per_b = (unsigned)per_b;
    if (nbR) hipLaunchKernelGGL(pt_row_assign_kernel, dim3(nbR), dim3(GDN_BLOCK), 0, 0, a.rowptr, mR, bcR, ro);
    hipLaunchKernelGGL(pt_ends_kernel, dim3(1), dim3(64), 0, 0, p.chunk_lo.p, nchunks, mL, p.bin_lo.p, nbins, mR, bin_e, crp,
                       (eoff_t)n_dst, (eoff_t)n, n_src0 ? 1 : 0, n_dst ? 1 : 0);
    GDN_HIP(hipGetLastError());
  }
  phase("ranks");
  // ---- the gather pass
  {
    PtKeyArgs ka;
    ka.colidx = a.colidx;
    ka.smap = smap_raw;
    ka.rowstart = reinterpret_cast<const unsigned long long *>(rowstart);
    ka.bin_e = bin_e;
    ka.crp = crp;
    ka.n_dst = (eoff_t)n_dst;
    ka.per_b = (unsigned)per_b;
    ka.nchunks = nchunks;
    ka.log_chunk = lc;
    ka.ntiers = ntiers;
    ka.S = S;
    ka.R = R;
    ka.tile_cnt = tile_cnt;
    ka.tier_cnt = tier_cnt;
    ka.errflag = errflag;
    hipLaunchKernelGGL(pt_keygen_kernel, dim3(nbins), dim3(PT_THREADS), (size_t)nchunks * 4, 0, ka);
    GDN_HIP(hipGetLastError());
  }
  phase("pt_keygen");
  // ---- offsets
  {
    const unsigned long long tt = ntiles > (unsigned long long)d1 * nbins ? ntiles : (unsigned long long)d1 * nbins;
    hipLaunchKernelGGL(pt_tile_sizes_kernel, dim3(gdn_nblocks(tt)), dim3(GDN_BLOCK), 0, 0, tile_cnt, nchunks, nbins, a.pad, d1, psz_c, psz_b, segsz);
    GDN_HIP(hipGetLastError());
    GDN_TRY(gdn_exclusive_scan_u32_to_u64_ws(psz_c, pu, (size_t)ntiles, ws, 0));
    GDN_TRY(gdn_exclusive_scan_u32_to_u64_ws(psz_b, pv, (size_t)ntiles, ws, 0));
    GDN_TRY(gdn_exclusive_scan_u32_to_u64_ws(segsz, segoff, (size_t)d1 * nbins, ws, 0));
    hipLaunchKernelGGL(pb_ptrs_kernel, dim3(gdn_nblocks((uint64_t)(nchunks > nbins ? nchunks : nbins) + 1)), dim3(GDN_BLOCK), 0, 0, pu,
                       pv, nchunks, nbins, p.chunk_ptr.p, p.bin_ptr.p);
    if (ntiers) {
      PtTierPads pads;
      for (int t = 0; t < PB_MAX_REC_TIERS; t++) {
        ts.t[t].interleaved = a.interleave && !(t == 0 && has_hub);
        pads.pad[t] = ts.t[t].interleaved ? 256u : 16u;
      }
      hipLaunchKernelGGL(pt_tier_sizes_kernel, dim3(gdn_nblocks(nbins)), dim3(GDN_BLOCK), 0, 0, tier_cnt, nbins, ntiers, tsz, pads);
      for (int t = 0; t < ntiers; t++) GDN_TRY(gdn_exclusive_scan_u32_to_u64_ws(tsz + (size_t)t * nbins, ts.t[t].bin_ptr.p, nbins, ws, 0));
    }
    hipLaunchKernelGGL(pt_sizes_from_ptr_kernel, dim3(gdn_nblocks(nbins)), dim3(GDN_BLOCK), 0, 0, bin_e, nbins, bin_sz);
    GDN_HIP(hipGetLastError());
  }
  std::vector<eoff_t> cs((size_t)nchunks + 1), bs((size_t)nbins + 1), h_binsz(nbins);
  std::vector<uint32_t> h_tier_cnt((size_t)nbins * 8);
  eoff_t n0 = 0, tier_pad[PB_MAX_REC_TIERS] = {};
  unsigned h_err = 0;
  GDN_HIP(hipMemcpy(cs.data(), p.chunk_ptr.p, cs.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
  GDN_HIP(hipMemcpy(bs.data(), p.bin_ptr.p, bs.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
  GDN_HIP(hipMemcpy(h_binsz.data(), bin_sz, (size_t)nbins * sizeof(eoff_t), hipMemcpyDeviceToHost));
  GDN_HIP(hipMemcpy(&n0, segoff + (size_t)d1 * nbins, sizeof(eoff_t), hipMemcpyDeviceToHost));
  GDN_HIP(hipMemcpy(&h_err, errflag, 4, hipMemcpyDeviceToHost));
  if (ntiers) GDN_HIP(hipMemcpy(h_tier_cnt.data(), tier_cnt, h_tier_cnt.size() * 4, hipMemcpyDeviceToHost));
  for (int t = 0; t < ntiers; t++)
    GDN_HIP(hipMemcpy(&tier_pad[t], ts.t[t].bin_ptr.p + nbins, sizeof(eoff_t), hipMemcpyDeviceToHost));
  if (trace) fprintf(stderr, "[pb_build_tiered]   main edges %llu, mismatch flag %u\n", (unsigned long long)n0, h_err);
  if (h_err) {
    gdn_set_error("layout build: a column id occurs whose source count is 0 (counts do not match the graph)");
    return 2;  // the caller repeats the build with exact marks
  }
  for (int t = 0; t < ntiers; t++) {
    uint64_t s = 0;
    for (unsigned b = 0; b < nbins; b++) s += h_tier_cnt[(size_t)b * 8 + t];
    ts.t[t].nnz = s;
  }
  p.nnz = n0;
  phase("offsets + readback");
  // slice starts on aligned boundaries (see pb_build)
  eoff_t n_pad = 0, al_b_used = 0;
  std::vector<eoff_t> ca((size_t)nchunks + 1, 0), ba((size_t)nbins + 1, 0);
  {
    std::vector<eoff_t> du(nchunks), dv(nbins);
    auto pick_align = [](eoff_t total, unsigned parts) {
      eoff_t al = 16;
      while (al < 16384 && al * 32 <= total / (parts ? parts : 1)) al <<= 1;
      return al;
    };
    const eoff_t al_c = pick_align(cs[nchunks], nchunks), al_b = pick_align(bs[nbins], nbins);
    al_b_used = al_b;
    for (unsigned c = 0; c < nchunks; c++) {
      du[c] = ca[c] - cs[c];
      ca[c + 1] = (ca[c] + (cs[c + 1] - cs[c]) + al_c - 1) & ~(al_c - 1);
    }
    for (unsigned b = 0; b < nbins; b++) {
      dv[b] = ba[b] - bs[b];
      ba[b + 1] = (ba[b] + (bs[b + 1] - bs[b]) + al_b - 1) & ~(al_b - 1);
    }
    n_pad = ca[nchunks] > ba[nbins] ? ca[nchunks] : ba[nbins];
    GDN_HIP(hipMemcpyAsync(d_du, du.data(), du.size() * sizeof(eoff_t), hipMemcpyHostToDevice, 0));
    GDN_HIP(hipMemcpyAsync(d_dv, dv.data(), dv.size() * sizeof(eoff_t), hipMemcpyHostToDevice, 0));
    hipLaunchKernelGGL(pb_shift_kernel, dim3(gdn_nblocks(ntiles)), dim3(GDN_BLOCK), 0, 0, pu, pv, d_du, d_dv, nchunks, nbins);
    GDN_HIP(hipMemcpyAsync(p.chunk_ptr.p, ca.data(), ca.size() * sizeof(eoff_t), hipMemcpyHostToDevice, 0));
    GDN_HIP(hipMemcpyAsync(p.bin_ptr.p, ba.data(), ba.size() * sizeof(eoff_t), hipMemcpyHostToDevice, 0));
    GDN_HIP(hipStreamSynchronize(0));  // du / dv are host vectors of this scope
  }
  p.n_pad = n_pad;
  if ((n_pad >> a.log_group) + 1 > 0xFFFFFFFFull) {
    gdn_set_error("layout build: more than 2^35 padded edges");
    return GDN_ERR_INVALID;
  }
  // bins by descending edge count: the launch order of the per-bin kernels
  {
    std::vector<uint32_t> bo(nbins);
    for (unsigned b = 0; b < nbins; b++) bo[b] = b;
    std::stable_sort(bo.begin(), bo.end(), [&](uint32_t x, uint32_t y) { return h_binsz[x] > h_binsz[y]; });
    GDN_HIP(hipMemcpy(order, bo.data(), (size_t)nbins * 4, hipMemcpyHostToDevice));
  }
  // ---- arena 3: X = main-layout items + the tiers' records in CSR order; the radix passes ping-pong with S's memory
  eoff_t tier_base[PB_MAX_REC_TIERS], xlen = n0;
  for (int t = 0; t < ntiers; t++) {
    tier_base[t] = xlen;
    xlen += tier_pad[t];
  }
  const bool with_vals = a.edge_vals != nullptr;
  GDN_REQUIRE(!with_vals || a.main_vals, "layout build: edge values without a place for the main layout's");
  PtArena A3;
  GDN_TRY(A3.init((pt_pad256((size_t)xlen * 4) + pt_pad256((size_t)(xlen - n0) * 4)) * (with_vals ? 2 : 1) + 4096, 4));
  uint32_t *X = A3.get<uint32_t>(xlen);
  uint32_t *TMP = A3.get<uint32_t>(xlen - n0);  // second buffer of the tiers' radix passes (laid out like X's tier part)
  uint32_t *XV = with_vals ? A3.get<uint32_t>(xlen) : nullptr;         // the items' value words, laid out like X
  uint32_t *TMPV = with_vals ? A3.get<uint32_t>(xlen - n0) : nullptr;  //   and like TMP
  PT_CHECK_PTR(X);
  PT_CHECK_PTR(TMP);
  if (with_vals) {
    PT_CHECK_PTR(XV);
    PT_CHECK_PTR(TMPV);
    GDN_TRY(a.main_vals->alloc(n_pad + grp));
    GDN_HIP(hipMemsetAsync(a.main_vals->p, 0, (n_pad + grp) * sizeof(float), 0));
    for (int t = 0; t < ntiers; t++) GDN_TRY(ts.t[t].A.alloc(tier_pad[t] + 16));
  }
  GDN_TRY(p.U.alloc(n_pad + grp));
  GDN_TRY(p.V.alloc(n_pad + grp));
  GDN_TRY(p.G.alloc((n_pad >> a.log_group) + 1));
  for (int t = 0; t < ntiers; t++) GDN_TRY(ts.t[t].rec.alloc(tier_pad[t] + 16));
  {
    const unsigned long long fb = (n_pad + grp + GDN_BLOCK - 1) / GDN_BLOCK;
    hipLaunchKernelGGL(pb_fill_u16_kernel, dim3((unsigned)(fb > 262144ull ? 262144ull : fb)), dim3(GDN_BLOCK), 0, 0, p.U.p, n_pad + grp,
                       (uint16_t)p.chunk_slots);
    GDN_HIP(hipMemsetAsync(p.V.p, 0, (n_pad + grp) * sizeof(uint16_t), 0));
    const unsigned long long ng = (n_pad >> a.log_group) + 1;
    const unsigned long long fbg = (ng + GDN_BLOCK - 1) / GDN_BLOCK;
    hipLaunchKernelGGL(pb_fill_u32_kernel, dim3((unsigned)(fbg > 262144ull ? 262144ull : fbg)), dim3(GDN_BLOCK), 0, 0, p.G.p, ng,
                       (uint32_t)(n_pad >> a.log_group));
  }
  phase("arena 3 + U V G rec + fills");
  int dbits = 1;
  while ((1u << dbits) < nd_split) dbits++;
  {
    PtSplitArgs sp;
    memset(&sp, 0, sizeof(sp));
    sp.S = S;
    sp.R = R;
    sp.bin_e = bin_e;
    sp.segoff = segoff;
    for (int t = 0; t < ntiers; t++) {
      sp.tier_ptr[t] = ts.t[t].bin_ptr.p;
      sp.tier_base[t] = tier_base[t];
    }
    sp.order = order;
    sp.X = X;
    sp.EV = reinterpret_cast<const uint32_t *>(a.edge_vals);
    sp.XV = XV;
    sp.d1 = d1;
    sp.ntiers = ntiers;
    sp.log_chunk = lc;
    sp.dbits = dbits;
    const size_t lds = pt_stage_bytes(nd_split, with_vals);
    auto *const kern = with_vals ? (stable ? &pt_split_kernel<true, true> : &pt_split_kernel<false, true>)
                                 : (stable ? &pt_split_kernel<true, false> : &pt_split_kernel<false, false>);
    GDN_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (n) hipLaunchKernelGGL(kern, dim3(nbins), dim3(PT_PTHREADS), lds, 0, sp);
    GDN_HIP(hipGetLastError());
  }
  phase("pt_split");
  if (n0) {
    const unsigned long long nseg = (unsigned long long)d1 * nbins;
    hipLaunchKernelGGL(pt_tiles_kernel, dim3(gdn_nblocks(nseg * 64)), dim3(GDN_BLOCK), 0, 0, X, segoff, d1, nbins, nchunks, pu, pv, p.U.p, p.V.p,
                       (const uint32_t *)XV, with_vals ? reinterpret_cast<uint32_t *>(a.main_vals->p) : nullptr);
    GDN_HIP(hipGetLastError());
  }
  p.v_il = false;
  if (a.v_interleave && al_b_used >= 512 && ba[nbins] >= 512) {
    // every bin starts (and ends) on a multiple of 512 edges: V in lane-interleaved blocks (phase B: one 16-byte load = the rows
    // of the lane's two quads of a block)
    const unsigned long long nblk = (unsigned long long)ba[nbins] >> 9;
    const unsigned long long wb = (nblk + GDN_WAVES_PER_BLOCK - 1) / GDN_WAVES_PER_BLOCK;
    hipLaunchKernelGGL(pb_v_interleave_kernel, dim3((unsigned)(wb > 65536ull ? 65536ull : wb)), dim3(GDN_BLOCK), 0, 0, p.V.p, nblk);
    GDN_HIP(hipGetLastError());
    p.v_il = true;
  }
  phase("pt_tiles");
  // record tiers: (row, source) -> (source, row) inside every bin
  auto *const kern_r = with_vals ? (stable ? &pt_radix_kernel<true, true> : &pt_radix_kernel<false, true>)
                                 : (stable ? &pt_radix_kernel<true, false> : &pt_radix_kernel<false, false>);
  if (ntiers) GDN_HIP(hipFuncSetAttribute((const void *)kern_r, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pt_stage_bytes(PT_MAX_DIGITS, with_vals)));
  for (int t = 0; t < ntiers; t++) {
    int nbits = 1;
    while ((1u << nbits) < ts.t[t].n_src) nbits++;
    const int passes = nbits > 10 ? 2 : 1;
    const int b1 = passes == 2 ? (nbits + 1) / 2 : nbits;
    uint32_t *tmp = TMP + (tier_base[t] - n0);
    uint32_t *tmpv = with_vals ? TMPV + (tier_base[t] - n0) : nullptr;
    PtRadixArgs ra;
    ra.ptr = ts.t[t].bin_ptr.p;
    ra.tier_cnt = tier_cnt;
    ra.order = order;
    ra.tier = t;
    ra.zrec = ts.t[t].n_src << PB_MID_ROW_BITS;
    for (int ps = 0; ps < passes; ps++) {
      ra.in = ps == 0 ? X + tier_base[t] : tmp;
      ra.out = ps == passes - 1 ? ts.t[t].rec.p : tmp;
      ra.vin = with_vals ? (ps == 0 ? XV + tier_base[t] : tmpv) : nullptr;
      ra.vout = with_vals ? (ps == passes - 1 ? reinterpret_cast<uint32_t *>(ts.t[t].A.p) : tmpv) : nullptr;
      ra.shift = PB_MID_ROW_BITS + (ps == 0 ? 0 : b1);
      ra.bits = ps == 0 ? b1 : nbits - b1;
      ra.last = ps == passes - 1 ? 1 : 0;
      const size_t lds = pt_stage_bytes(1u << ra.bits, with_vals);
      hipLaunchKernelGGL(kern_r, dim3(nbins), dim3(PT_PTHREADS), lds, 0, ra);
    }
    if (ts.t[t].interleaved && tier_pad[t]) {
      const unsigned long long nblk = (unsigned long long)tier_pad[t] >> 8;  // every stream is a whole number of blocks
      const unsigned long long wb = (nblk + GDN_WAVES_PER_BLOCK - 1) / GDN_WAVES_PER_BLOCK;
      hipLaunchKernelGGL(pt_interleave_kernel, dim3((unsigned)(wb > 65536ull ? 65536ull : wb)), dim3(GDN_BLOCK), 0, 0, ts.t[t].rec.p, nblk);
      if (with_vals)  // the values in the same interleaved order
        hipLaunchKernelGGL(pt_interleave_kernel, dim3((unsigned)(wb > 65536ull ? 65536ull : wb)), dim3(GDN_BLOCK), 0, 0,
                           reinterpret_cast<uint32_t *>(ts.t[t].A.p), nblk);
    }
    GDN_HIP(hipGetLastError());
  }
  phase("pt_radix");
  hipLaunchKernelGGL(pb_groups_kernel, dim3(gdn_nblocks(ntiles)), dim3(GDN_BLOCK), 0, 0, pu, pv, psz_c, nchunks, nbins, p.G.p, 0, a.log_group);
  GDN_HIP(hipGetLastError());
  {  // largest-first launch order of both phases
    std::vector<uint32_t> co(nchunks), bo(nbins);
    for (unsigned i = 0; i < nchunks; i++) co[i] = i;
    for (unsigned i = 0; i < nbins; i++) bo[i] = i;
    std::stable_sort(co.begin(), co.end(), [&](uint32_t x, uint32_t y) { return ca[x + 1] - ca[x] > ca[y + 1] - ca[y]; });
    std::stable_sort(bo.begin(), bo.end(), [&](uint32_t x, uint32_t y) { return ba[x + 1] - ba[x] > ba[y + 1] - ba[y]; });
    GDN_TRY(p.chunk_order.alloc(nchunks));
    GDN_TRY(p.bin_order.alloc(nbins));
    GDN_HIP(hipMemcpyAsync(p.chunk_order.p, co.data(), co.size() * 4, hipMemcpyHostToDevice, 0));
    GDN_HIP(hipMemcpyAsync(p.bin_order.p, bo.data(), bo.size() * 4, hipMemcpyHostToDevice, 0));
    GDN_HIP(hipStreamSynchronize(0));
  }
  if (a.alloc_vals) {
    GDN_TRY(p.vals.alloc(n_pad + grp));
    GDN_HIP(hipMemsetAsync(p.vals.p, 0, (n_pad + grp) * sizeof(float), 0));
  }
  GDN_TRY(p.partial.alloc(nbins));
  GDN_TRY(p.red_scratch.alloc(2 * ((size_t)nbins / 4096 + 2)));
  ts.n = ntiers;
  ts.first_is_hub = has_hub && ntiers > 0;
  GDN_HIP(hipDeviceSynchronize());
  phase("groups + orders + vals");
  if (trace) {
    fprintf(stderr, "[pb_build_tiered] edges %llu: main %llu padded %llu (%.3f x) chunks %u (%llu slots) bins %u", n,
            (unsigned long long)n0, (unsigned long long)n_pad, n0 ? (double)n_pad / (double)n0 : 0.0, nchunks,
            (unsigned long long)per_c, nbins);
    for (int t = 0; t < ntiers; t++) fprintf(stderr, ", tier %d: %u sources %llu edges", t, ts.t[t].n_src, (unsigned long long)ts.t[t].nnz);
    fprintf(stderr, "; %.2f ms wall\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  }
  return GDN_OK;
}

// =====================================================================================================================
// OUT-CSR orientation (SSSP's dense sweeps, gdn_sssp.hip): rows = SOURCES, columns = destinations, no vertex compaction,
// one 32-bit value per edge (the weight) travels with it.  The same two-level split with the roles swapped -- a workgroup
// per source CHUNK (a contiguous edge range of an out-CSR), digits = destination bins -- and the record tiers fall out of
// the same passes: CSR order is source-major already, so the records of a (tier, bin) arrive sorted by source and need
// no radix pass.  Replaces pb_build(rows_are_sources) + sssp_build_tiers: a full 7-pass sort of 8-byte keys, a binary
// search per edge for its weight and a second pair of sorts for the tiers (RMAT-24: 25 ms of preparation).
//   items   main layout: slot << 17 | row << 2 | bin & 3, value = weight          (level-1 digit: bin >> 2)
//           tier t:      index << 15 | row,               value = weight & 255 | (bin & 31) << 8   (digit: D1 + t * DT + bin >> 5)
// =====================================================================================================================
#define PO_LOW 2    // bin bits left to the tile pass (slot 15 + row 15 + 2)
#define PO_TLOW 5   // bin bits left to the tier pass (they ride in the value word)
#define PO_ROW_BITS 15

// per-row codes of an out-CSR: class << 29 | (class ? index in the tier : the row id itself)
static __global__ void __launch_bounds__(GDN_BLOCK)
po_src_assign_kernel(PtSrcArgs a, unsigned nb, const uint32_t *__restrict__ bc, uint32_t *__restrict__ smap, PtSrcOut o) {
  __shared__ unsigned long long s_scan[GDN_WAVES_PER_BLOCK];
  const unsigned base = blockIdx.x * PT_VTILE + threadIdx.x * 8u;
  int cls[8];
  unsigned long long pa = 0ull, pb = 0ull;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const unsigned s = base + (unsigned)i;
    cls[i] = s < a.n ? pt_class_of(a, s) : PT_NCLS;
    if (cls[i] < 3) pa += 1ull << (16 * cls[i]);
    else if (cls[i] < PT_NCLS) pb += 1ull << (16 * (cls[i] - 3));
  }
  unsigned long long ta, tb;
  const unsigned long long ea = gdn_block_excl_scan(pa, s_scan, &ta);
  __syncthreads();
  const unsigned long long eb = gdn_block_excl_scan(pb, s_scan, &tb);
  unsigned run[PT_NCLS];
#pragma unroll
  for (int k = 0; k < PT_NCLS; k++) {
    const unsigned long long e = k < 3 ? ea : eb;
    run[k] = bc[(size_t)k * nb + blockIdx.x] + (unsigned)((e >> (16 * (k % 3))) & 0xFFFFull);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const unsigned s = base + (unsigned)i;
    if (s >= a.n) continue;
    const int k = cls[i];
    unsigned code = s;  // class 0 (and rows without edges: never read)
    if (k >= 1 && k < PT_NCLS) {
      unsigned r = 0;
#pragma unroll
      for (int q = 1; q < PT_NCLS; q++)
        if (k == q) r = run[q]++;
      code = ((unsigned)k << PT_CLASS_SHIFT) | r;
      o.ids[k - 1][r] = s;
    }
    smap[s] = code;
  }
}

// degrees as the count array of the histogram / class kernels
static __global__ void __launch_bounds__(GDN_BLOCK)
po_degrees_kernel(const eoff_t *__restrict__ rowptr, unsigned m, int32_t *__restrict__ deg, uint32_t *__restrict__ cnt) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < m) {
    const eoff_t d = rowptr[i + 1] - rowptr[i];
    const uint32_t c = d > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)d;
    deg[i] = (int32_t)c;
    cnt[i] = c;
  }
}

// rows with entries: compact list (row id and first edge of the k-th one), row-start bitmap, and for every chunk the
// number of such rows in front of it
static __global__ void __launch_bounds__(GDN_BLOCK)
po_row_assign_kernel(const eoff_t *__restrict__ rowptr, unsigned m, const uint32_t *__restrict__ bc, int log_chunk,
                     eoff_t *__restrict__ crp, uint32_t *__restrict__ nzrow, uint32_t *__restrict__ rowstart,
                     uint32_t *__restrict__ chunk_k) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK];
  const unsigned base = blockIdx.x * PT_VTILE + threadIdx.x * 8u;
  eoff_t st[9];
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const unsigned r = base + (unsigned)i;
    st[i] = r <= m ? rowptr[r] : 0;
  }
  unsigned act = 0u;
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (base + (unsigned)i < m && st[i + 1] > st[i]) {
      act |= 1u << i;
      c++;
    }
  unsigned total;
  unsigned k = bc[blockIdx.x] + gdn_block_excl_scan(c, s_scan, &total);
  const unsigned cmask = (1u << log_chunk) - 1u;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const unsigned r = base + (unsigned)i;
    if (r < m && (r & cmask) == 0u) chunk_k[r >> log_chunk] = k;
    if (!((act >> i) & 1u)) continue;
    crp[k] = st[i];
    nzrow[k] = r;
    atomicOr(&rowstart[st[i] >> 5], 1u << (st[i] & 31u));
    k++;
  }
}

// pass 1 over the edges, a workgroup per source chunk: S[e] = code of the edge's source, counts per (class, bin)
struct PoCountArgs {
  const eoff_t *rowptr;
  const vid_t *colidx;
  const uint32_t *smap;
  const unsigned long long *rowstart;
  const eoff_t *crp;
  const uint32_t *nzrow, *chunk_k;
  eoff_t n_nz;
  unsigned m, nchunks, nbins, ncls;
  int log_chunk, log_bin;
  uint32_t *S;
  uint32_t *cnt;  // ncls x nchunks x nbins
};
static __global__ void __launch_bounds__(PT_THREADS)
po_count_kernel(PoCountArgs a) {
  extern __shared__ unsigned s_cnt[];  // ncls x nbins
  const unsigned c = blockIdx.x;
  const unsigned nc = a.ncls * a.nbins;
  for (unsigned i = threadIdx.x; i < nc; i += PT_THREADS) s_cnt[i] = 0u;
  __syncthreads();
  const unsigned r0 = c << a.log_chunk, r1 = (c + 1) << a.log_chunk < a.m ? (c + 1) << a.log_chunk : a.m;
  const eoff_t E0 = a.rowptr[r0], E1 = a.rowptr[r1];
  const unsigned w = threadIdx.x >> 6, lane = gdn_lane();
  if (E1 > E0) {
    const eoff_t w0 = E0 >> 6, w1 = (E1 - 1) >> 6;
    constexpr unsigned KW = PT_THREADS / 64;
    const eoff_t per = (w1 - w0 + KW) / KW;
    const eoff_t wa = w0 + (eoff_t)w * per, wb = wa + per < w1 + 1 ? wa + per : w1 + 1;
    if (wa < wb) {
      const eoff_t kb0 = a.chunk_k[c], kb1 = c + 1 < a.nchunks ? a.chunk_k[c + 1] : a.n_nz;
      eoff_t x = wa << 6;
      if (x < E0) x = E0;
      eoff_t lo = kb0, hi = kb1;
      while (lo < hi) {
        const eoff_t mid = lo + ((hi - lo) >> 1);
        if (a.crp[mid] < x) lo = mid + 1;
        else hi = mid;
      }
      eoff_t rb = lo;
      const unsigned long long le = lane == 63u ? ~0ull : ((2ull << lane) - 1ull);
      constexpr int UNR = 4;
      for (eoff_t wi = wa; wi < wb; wi += UNR) {
        unsigned long long bits[UNR];
        vid_t col[UNR];
        bool ok[UNR];
#pragma unroll
        for (int r = 0; r < UNR; r++) {
          const eoff_t wj = wi + r, e = (wj << 6) + lane;
          ok[r] = wj < wb && e >= E0 && e < E1;
          bits[r] = wj < wb ? a.rowstart[wj] : 0ull;
          col[r] = ok[r] ? __builtin_nontemporal_load(a.colidx + e) : 0;
        }
#pragma unroll
        for (int r = 0; r < UNR; r++) {
          const eoff_t wj = wi + r;
          if (wj >= wb) break;
          unsigned long long bt = bits[r];
          if ((wj << 6) < E0) bt &= ~0ull << (E0 & 63u);
          if (((wj + 1) << 6) > E1) bt &= ~0ull >> (64u - (unsigned)(E1 & 63u));
          const eoff_t k = rb + (eoff_t)__popcll(bt & le) - 1;
          rb += (eoff_t)__popcll(bt);
          const eoff_t e = (wj << 6) + lane;
          uint32_t code = 0u;
          if (ok[r]) {
            code = a.smap[a.nzrow[k]];
            a.S[e] = code;
          }
          // one LDS atomic per run of equal (class, bin) in the wave
          const unsigned key = ok[r] ? (code >> PT_CLASS_SHIFT) * a.nbins + ((unsigned)col[r] >> a.log_bin) : 0xFFFFFFFFu;
          const unsigned prev = (unsigned)__shfl_up((int)key, 1, 64);
          const bool head = lane == 0u || prev != key;
          const unsigned long long heads = __ballot(head);
          if (head && key != 0xFFFFFFFFu) {
            const unsigned long long after = lane == 63u ? 0ull : heads >> (lane + 1u);
            const unsigned len = after ? (unsigned)__ffsll((long long)after) : 64u - lane;
            atomicAdd(&s_cnt[key], len);
          }
        }
      }
    }
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < nc; i += PT_THREADS) {
    const unsigned k = i / a.nbins, b = i - k * a.nbins;
    a.cnt[((size_t)k * a.nchunks + c) * a.nbins + b] = s_cnt[i];
  }
}

// offsets: padded tile sizes of the main layout in both orders, segment sizes of the split's output (per chunk: D1 main
// digits, then DT per tier), and per (tier, bin) the exclusive prefix over the chunks
static __global__ void __launch_bounds__(GDN_BLOCK)
po_sizes_kernel(const uint32_t *__restrict__ cnt, unsigned nchunks, unsigned nbins, unsigned ntiers, unsigned pad, unsigned d1, unsigned dt,
                uint32_t *__restrict__ psz_c, uint32_t *__restrict__ psz_b, uint32_t *__restrict__ segsz,
                const uint32_t *__restrict__ tsz /*ntiers x nbins totals: the tier planes of cnt hold PREFIXES over the chunks*/) {
  const unsigned long long t = (unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (t < (unsigned long long)nchunks * nbins) {  // chunk-major tile index
    const unsigned c = (unsigned)(t / nbins), b = (unsigned)(t % nbins);
    const uint32_t sz = (cnt[t] + (pad - 1u)) & ~(pad - 1u);
    psz_c[t] = sz;
    psz_b[(unsigned long long)b * nchunks + c] = sz;
  }
  const unsigned nd = d1 + ntiers * dt;
  if (t < (unsigned long long)nchunks * nd) {
    const unsigned c = (unsigned)(t / nd), d = (unsigned)(t % nd);
    uint32_t s = 0;
    if (d < d1) {
      for (unsigned k = 0; k < (1u << PO_LOW); k++) {
        const unsigned b = (d << PO_LOW) + k;
        if (b < nbins) s += cnt[(unsigned long long)c * nbins + b];
      }
    } else {
      const unsigned tt = (d - d1) / dt, h = (d - d1) % dt;
      for (unsigned k = 0; k < (1u << PO_TLOW); k++) {
        const unsigned b = (h << PO_TLOW) + k;
        if (b < nbins) {
          const uint32_t here = cnt[((unsigned long long)(1 + tt) * nchunks + c) * nbins + b];
          const uint32_t next = c + 1 < nchunks ? cnt[((unsigned long long)(1 + tt) * nchunks + c + 1) * nbins + b] : tsz[(size_t)tt * nbins + b];
          s += next - here;
        }
      }
    }
    segsz[t] = s;
  }
}
static __global__ void __launch_bounds__(GDN_BLOCK)
po_tier_prefix_kernel(uint32_t *__restrict__ cnt, unsigned nchunks, unsigned nbins, unsigned ntiers, uint32_t *__restrict__ tsz) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;  // (tier, bin)
  if (i >= ntiers * nbins) return;
  const unsigned t = i / nbins, b = i - t * nbins;
  uint32_t acc = 0;
  for (unsigned c = 0; c < nchunks; c++) {
    uint32_t *p = cnt + ((unsigned long long)(1 + t) * nchunks + c) * nbins + b;
    const uint32_t v = *p;
    *p = acc;
    acc += v;
  }
  tsz[i] = acc;
}

// interleaved streams of an out-CSR plan: padded stream sizes, and the in-place interleave of the whole 256-record blocks of
// every (tier, bin) stream -- records as in pt_interleave_kernel, the weight bytes with the same permutation (a lane's
// 32-bit word = the weights of its four records).  A workgroup per stream, a wave per block.
static __global__ void __launch_bounds__(GDN_BLOCK)
po_pad_sizes_kernel(const uint32_t *__restrict__ tsz, unsigned n, uint32_t *__restrict__ out) {
  const unsigned i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i < n) out[i] = (tsz[i] + 255u) & ~255u;
}
static __global__ void __launch_bounds__(GDN_BLOCK)
po_interleave_kernel(uint32_t *__restrict__ rec, uint8_t *__restrict__ w8, const eoff_t *__restrict__ ptr, const uint32_t *__restrict__ cnt) {
  typedef unsigned pt_u32x4 __attribute__((ext_vector_type(4)));
  const unsigned lane = gdn_lane();
  const eoff_t j0 = ptr[blockIdx.x];
  const unsigned nblk = cnt[blockIdx.x] >> 8;
  for (unsigned q = threadIdx.x >> 6; q < nblk; q += GDN_WAVES_PER_BLOCK) {
    uint32_t *base = rec + j0 + (eoff_t)q * 256u;
    pt_u32x4 v;
    v.x = base[lane];
    v.y = base[64u + lane];
    v.z = base[128u + lane];
    v.w = base[192u + lane];
    unsigned wv = 0u;
    uint8_t *wb = w8 ? w8 + j0 + (eoff_t)q * 256u : nullptr;
    if (wb) wv = (unsigned)wb[lane] | ((unsigned)wb[64u + lane] << 8) | ((unsigned)wb[128u + lane] << 16) | ((unsigned)wb[192u + lane] << 24);
    __builtin_amdgcn_s_waitcnt(0);  // every load of the wave has returned before its stores issue
    __builtin_amdgcn_wave_barrier();
    reinterpret_cast<pt_u32x4 *>(base)[lane] = v;
    if (wb) reinterpret_cast<uint32_t *>(wb)[lane] = wv;
  }
}

// (item, value) form of pt_partition: ranks by LDS atomics, both words staged
struct PoStage {
  uint32_t *stage, *vstage;
  unsigned short *dig;
  unsigned *wc, *start, *tot, *scr;
  unsigned long long *goff;
};
__device__ __forceinline__ PoStage po_stage_carve(unsigned char *lds, unsigned nd) {
  PoStage s;
  s.stage = reinterpret_cast<uint32_t *>(lds);
  s.vstage = s.stage + PT_STEP;
  s.goff = reinterpret_cast<unsigned long long *>(lds + (size_t)PT_STEP * 8);
  s.start = reinterpret_cast<unsigned *>(s.goff + nd);
  s.tot = s.start + nd;
  s.scr = s.tot + nd;
  s.wc = s.scr + PT_WAVES + 2;
  s.dig = reinterpret_cast<unsigned short *>(s.wc + (size_t)PT_WAVES * nd);
  return s;
}
static inline size_t po_stage_bytes(unsigned nd) {
  return (size_t)PT_STEP * 8 + (size_t)nd * 16 + (PT_WAVES + 2) * 4 + (size_t)PT_WAVES * nd * 4 + (size_t)PT_STEP * 2 + 64;
}
template <class Load>
__device__ __forceinline__ void po_partition(unsigned long long n, unsigned nd, uint32_t *__restrict__ out, uint32_t *__restrict__ out_v,
                                             const PoStage &st, Load load) {
  const unsigned w = threadIdx.x >> 6, lane = gdn_lane();
  unsigned *wcw = st.wc + (size_t)w * nd;
  constexpr int IPT = PT_IPT / 2;  // two words per item: eight items per thread and half-step
  for (unsigned long long base = 0; base < n; base += PT_STEP / 2) {
    const unsigned cnt = (unsigned)(n - base < (unsigned long long)(PT_STEP / 2) ? n - base : (unsigned long long)(PT_STEP / 2));
    uint32_t it[IPT], vv[IPT];
    unsigned short dg[IPT], rk[IPT];
#pragma unroll
    for (int j = 0; j < IPT; j++) {
      const unsigned i = w * (IPT * 64u) + (unsigned)j * 64u + lane;
      it[j] = 0u;
      vv[j] = 0u;
      unsigned d = 0u;
      if (i < cnt) load(base + i, it[j], vv[j], d);
      dg[j] = (unsigned short)d;
    }
#pragma unroll
    for (int j = 0; j < IPT; j++) {
      const unsigned i = w * (IPT * 64u) + (unsigned)j * 64u + lane;
      rk[j] = i < cnt ? (unsigned short)atomicAdd(&wcw[dg[j]], 1u) : (unsigned short)0;
    }
    __syncthreads();
    unsigned mine[2] = {0u, 0u};
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const unsigned d = 2u * threadIdx.x + (unsigned)q;
      if (d < nd) {
        unsigned acc = 0u;
#pragma unroll
        for (int ww = 0; ww < PT_WAVES; ww++) {
          const unsigned c = st.wc[(size_t)ww * nd + d];
          st.wc[(size_t)ww * nd + d] = acc;
          acc += c;
        }
        mine[q] = acc;
        st.tot[d] = acc;
      }
    }
    unsigned total;
    const unsigned ex = pt_block_excl_scan(mine[0] + mine[1], st.scr, &total);
    if (2u * threadIdx.x < nd) st.start[2u * threadIdx.x] = ex;
    if (2u * threadIdx.x + 1u < nd) st.start[2u * threadIdx.x + 1u] = ex + mine[0];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; j++) {
      const unsigned i = w * (IPT * 64u) + (unsigned)j * 64u + lane;
      if (i < cnt) {
        const unsigned d = dg[j];
        const unsigned pos = st.start[d] + wcw[d] + rk[j];
        st.stage[pos] = it[j];
        st.vstage[pos] = vv[j];
        st.dig[pos] = (unsigned short)d;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; j++) {
      const unsigned pos = (unsigned)j * PT_PTHREADS + threadIdx.x;
      if (pos < cnt) {
        const unsigned d = st.dig[pos];
        const unsigned long long at = st.goff[d] + (pos - st.start[d]);
        out[at] = st.stage[pos];
        out_v[at] = st.vstage[pos];
      }
    }
    __syncthreads();
    for (unsigned d = threadIdx.x; d < nd; d += PT_PTHREADS) {
      st.goff[d] += st.tot[d];
#pragma unroll
      for (int ww = 0; ww < PT_WAVES; ww++) st.wc[(size_t)ww * nd + d] = 0;
    }
    __syncthreads();
  }
}

struct PoSplitArgs {
  const eoff_t *rowptr;
  const vid_t *colidx;
  const uint32_t *S;
  const int32_t *weight;
  const eoff_t *segoff;  // nchunks x nd (+1)
  const uint32_t *order;
  uint32_t *X, *XV;
  unsigned m, d1, dt, ntiers;
  int log_chunk, log_bin;
};
static __global__ void __launch_bounds__(PT_PTHREADS, 4)
po_split_kernel(PoSplitArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const unsigned nd = a.d1 + a.ntiers * a.dt;
  const PoStage st = po_stage_carve(s_raw, nd);
  const unsigned c = a.order[blockIdx.x];
  for (unsigned d = threadIdx.x; d < nd; d += PT_PTHREADS) {
    st.goff[d] = a.segoff[(size_t)c * nd + d];
#pragma unroll
    for (int ww = 0; ww < PT_WAVES; ww++) st.wc[(size_t)ww * nd + d] = 0;
  }
  __syncthreads();
  const unsigned r0 = c << a.log_chunk, r1 = (c + 1) << a.log_chunk < a.m ? (c + 1) << a.log_chunk : a.m;
  const eoff_t E0 = a.rowptr[r0], E1 = a.rowptr[r1];
  const uint32_t *S = a.S + E0;
  const vid_t *col = a.colidx + E0;
  const int32_t *wt = a.weight + E0;
  const unsigned bmask = (1u << a.log_bin) - 1u, cmask = (1u << a.log_chunk) - 1u;
  const int lb = a.log_bin;
  const unsigned d1 = a.d1, dt = a.dt;
  po_partition(E1 - E0, nd, a.X, a.XV, st, [&](unsigned long long i, uint32_t &item, uint32_t &val, unsigned &d) {
    const uint32_t code = __builtin_nontemporal_load(S + i);
    const unsigned dst = (unsigned)__builtin_nontemporal_load(col + i);
    const unsigned wv = (unsigned)__builtin_nontemporal_load(wt + i);
    const unsigned k = code >> PT_CLASS_SHIFT, bin = dst >> lb, row = dst & bmask;
    if (k == 0u) {
      d = bin >> PO_LOW;
      item = ((code & cmask) << 17) | (row << PO_LOW) | (bin & ((1u << PO_LOW) - 1u));
      val = wv;
    } else {
      d = d1 + (k - 1u) * dt + (bin >> PO_TLOW);
      item = ((code & PT_IDX_MASK) << PO_ROW_BITS) | row;
      val = (wv & 0xFFu) | ((bin & ((1u << PO_TLOW) - 1u)) << 8);
    }
  });
}

// level 2, a wave per (chunk, digit) segment.  Main digits: split by the last 2 bin bits into U / W (contiguous) and V (one run per
// tile).  Tier digits: split by the last 5 bin bits into the (tier, bin) record streams behind the runs of the earlier chunks.
struct PoTilesArgs {
  const uint32_t *X, *XV;
  const eoff_t *segoff;
  const eoff_t *pu, *pv;       // main layout: tile offsets, chunk-major / bin-major
  const eoff_t *tptr;          // ntiers x nbins (+1) record offsets
  const uint32_t *tpre;        // (1 + t) x nchunks x nbins: records of (tier, bin) in the chunks in front (class 0 plane unused)
  uint16_t *U, *V;
  uint32_t *W;
  uint32_t *rec;
  uint8_t *w8;
  unsigned nchunks, nbins, d1, dt, ntiers;
};
static __global__ void __launch_bounds__(GDN_BLOCK)
po_tiles_kernel(PoTilesArgs a) {
  __shared__ unsigned s_run[GDN_WAVES_PER_BLOCK][1 << PO_TLOW];
  const unsigned nd = a.d1 + a.ntiers * a.dt;
  const unsigned long long seg = ((unsigned long long)blockIdx.x * GDN_BLOCK + threadIdx.x) >> 6;
  if (seg >= (unsigned long long)a.nchunks * nd) return;
  const unsigned c = (unsigned)(seg / nd), d = (unsigned)(seg % nd), lane = gdn_lane();
  const eoff_t s0 = a.segoff[seg], s1 = a.segoff[seg + 1];
  if (s1 == s0) return;
  const unsigned long long lt = gdn_lanemask_lt();
  if (d < a.d1) {
    eoff_t offu = 0, offv = 0;
    const unsigned bl = (d << PO_LOW) + (lane & ((1u << PO_LOW) - 1u));
    if (bl < a.nbins) {
      offu = a.pu[(unsigned long long)c * a.nbins + bl];
      offv = a.pv[(unsigned long long)bl * a.nchunks + c];
    }
    unsigned basek[1 << PO_LOW];
#pragma unroll
    for (int k = 0; k < (1 << PO_LOW); k++) basek[k] = 0u;
    constexpr int TU = 4;  // steps loaded ahead (a segment is a chain of dependent steps otherwise)
    for (eoff_t i0 = s0; i0 < s1; i0 += 64 * TU) {
      uint32_t items[TU], vals[TU];
#pragma unroll
      for (int q = 0; q < TU; q++) {
        const eoff_t i = i0 + (eoff_t)q * 64 + lane;
        items[q] = i < s1 ? __builtin_nontemporal_load(a.X + i) : 0u;
        vals[q] = i < s1 ? __builtin_nontemporal_load(a.XV + i) : 0u;
      }
#pragma unroll
      for (int q = 0; q < TU; q++) {
        const eoff_t i = i0 + (eoff_t)q * 64 + lane;
        if (i0 + (eoff_t)q * 64 >= s1) break;  // wave-uniform
        const bool valid = i < s1;
        const uint32_t item = items[q], val = vals[q];
        const unsigned low = item & ((1u << PO_LOW) - 1u);
        unsigned rank = 0u;
#pragma unroll
        for (int k = 0; k < (1 << PO_LOW); k++) {
          const unsigned long long mk = __ballot(valid && low == (unsigned)k);
          if (low == (unsigned)k) rank = basek[k] + (unsigned)__popcll(mk & lt);
          basek[k] += (unsigned)__popcll(mk);
        }
        const eoff_t ou = (eoff_t)__shfl((long long)offu, (int)low, 64), ov = (eoff_t)__shfl((long long)offv, (int)low, 64);
        if (valid) {
          a.U[ou + rank] = (uint16_t)(item >> 17);
          a.W[ou + rank] = val;
          a.V[ov + rank] = (uint16_t)((item >> PO_LOW) & ((1u << PO_ROW_BITS) - 1u));
        }
      }
    }
    return;
  }
  // a tier segment: up to 32 bins
  const unsigned t = (d - a.d1) / a.dt, h = (d - a.d1) % a.dt;
  eoff_t off = 0;
  const unsigned bl = (h << PO_TLOW) + (lane & ((1u << PO_TLOW) - 1u));
  if (bl < a.nbins)
    off = a.tptr[(unsigned long long)t * a.nbins + bl] + a.tpre[((unsigned long long)(1 + t) * a.nchunks + c) * a.nbins + bl];
  // records of bin k of the segment written so far: a counter per bin in the wave's LDS row, and a record's place in its bin
  // is ONE LDS atomic with return (round 4; a ballot per bin -- 32 of them per 64 records -- was most of this kernel's
  // instructions).  Lanes of one instruction that share a bin take their slots in the order the LDS unit serialises them
  // (lane order here, see pt_partition); nothing depends on it: the stream's order by source only keeps phase B's table
  // reads together.
  unsigned *run = s_run[threadIdx.x >> 6];
  if (lane < (1u << PO_TLOW)) run[lane] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  constexpr int TU = 4;
  for (eoff_t i0 = s0; i0 < s1; i0 += 64 * TU) {
    uint32_t items[TU], vals[TU];
#pragma unroll
    for (int q = 0; q < TU; q++) {
      const eoff_t i = i0 + (eoff_t)q * 64 + lane;
      items[q] = i < s1 ? __builtin_nontemporal_load(a.X + i) : 0u;
      vals[q] = i < s1 ? __builtin_nontemporal_load(a.XV + i) : 0u;
    }
#pragma unroll
    for (int q = 0; q < TU; q++) {
      const eoff_t i = i0 + (eoff_t)q * 64 + lane;
      if (i0 + (eoff_t)q * 64 >= s1) break;  // wave-uniform
      const bool valid = i < s1;
      const uint32_t item = items[q], val = vals[q];
      const unsigned low = (val >> 8) & ((1u << PO_TLOW) - 1u);
      const eoff_t o = (eoff_t)__shfl((long long)off, (int)low, 64);
      if (valid) {
        const unsigned before = atomicAdd(&run[low], 1u);
        a.rec[o + before] = item;
        if (a.w8) a.w8[o + before] = (uint8_t)(val & 0xFFu);
      }
    }
  }
}

// Tier thresholds of an out-CSR plan from the histograms of the out-degrees: tier t takes the sources with
// thr[t] <= degree < thr[t-1], at most caps[t] of them, none below min_deg.  Degrees from 4095 on sit in one open bin of
// the linear histogram (h_lin[4095]) -- a threshold up there is a quarter-octave bucket floor of h_log.
static int po_choose_tiers(const unsigned *h_log, const unsigned *h_lin, unsigned min_deg, int max_tiers, const unsigned *caps,
                           unsigned *thr) {
  // candidate thresholds, descending, with the number of sources at or above each
  std::vector<std::pair<unsigned, unsigned long long>> cand;
  {
    unsigned long long acc = 0;
    for (int bk = PB_HUB_BUCKETS - 1; bk >= 0; bk--) {
      acc += h_log[bk];
      const unsigned fl = pb_hub_bucket_floor((unsigned)bk);
      if (fl > PB_LIN_BINS - 1u && acc) cand.push_back(std::make_pair(fl, acc));
    }
    acc = 0;
    for (unsigned d = PB_LIN_BINS - 1u; d >= 1u; d--) {
      acc += h_lin[d];
      if (d >= min_deg && acc) cand.push_back(std::make_pair(d, acc));
    }
  }
  int nt = 0;
  unsigned long long above = 0;
  size_t i = 0;
  for (int t = 0; t < max_tiers && i < cand.size(); t++) {
    size_t best = cand.size();
    while (i < cand.size() && cand[i].second - above <= caps[t]) best = i++;
    if (best == cand.size() || cand[best].second == above) break;  // nothing fits (one degree holds more sources than a tier)
    thr[nt++] = cand[best].first;
    above = cand[best].second;
  }
  return nt;
}

// GDN_OK; 1 = outside the limits (nothing built)
static int pb_build_out_tiered(const PbOutArgs &a, PbPlan &p, DevBuf<float> &Wp, PbOutTiers &ts) {
  const auto t_begin = std::chrono::steady_clock::now();
  const bool trace = gdn_xoption("GDN_PB_TRACE") != nullptr;
  auto t_last = t_begin;
  auto phase = [&](const char *name) {
    if (!trace) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[pb_build_out]   %-24s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  const gdn_graph *g = a.g;
  const unsigned m = (unsigned)g->m;
  const unsigned long long n = g->nnz;
  const int lc = a.log_chunk, lb = a.log_bin;
  GDN_REQUIRE(lc >= 8 && lc <= 15 && lb >= 8 && lb <= PO_ROW_BITS, "out layout: log_chunk / log_bin");
  const unsigned nchunks = (m + (1u << lc) - 1) >> lc, nbins = (m + (1u << lb) - 1) >> lb;
  const unsigned d1 = (nbins + (1u << PO_LOW) - 1) >> PO_LOW, dt = (nbins + (1u << PO_TLOW) - 1) >> PO_TLOW;
  int max_tiers = a.max_tiers > PB_MAX_REC_TIERS ? PB_MAX_REC_TIERS : a.max_tiers;
  if (d1 + (unsigned)max_tiers * dt > PT_MAX_DIGITS || nchunks == 0 || nbins > 8192 || n == 0) return 1;
  const unsigned grp = 1u << a.log_group;
  const unsigned nbV = (m + PT_VTILE - 1) / PT_VTILE;
  ts.n = 0;
  ts.edges = 0;
  PtArena A1;
  {
    size_t bytes = pt_pad256((size_t)m * 4) * 3 + pt_pad256((size_t)PT_NCLS * nbV * 4) + pt_pad256((size_t)nbV * 4) + 8192;
    bytes += pt_pad256(((size_t)m + 2) * 8) + pt_pad256(((size_t)m + 2) * 4) + pt_pad256(((size_t)nchunks + 2) * 4);
    bytes += pt_pad256(((size_t)(n >> 5) + 4) * 4) + pt_pad256((size_t)n * 4);
    bytes += pt_pad256((PB_HUB_BUCKETS + PB_LIN_BINS + 64) * 4);
    GDN_TRY(A1.init(bytes));
  }
  int32_t *deg = A1.get<int32_t>(m);
  uint32_t *cnt16 = A1.get<uint32_t>(m), *smap = A1.get<uint32_t>(m);
  uint32_t *bcS = A1.get<uint32_t>((size_t)PT_NCLS * nbV), *bcR = A1.get<uint32_t>(nbV), *totals = A1.get<uint32_t>(16);
  unsigned *hist = A1.get<unsigned>(PB_HUB_BUCKETS + PB_LIN_BINS);
  eoff_t *crp = A1.get<eoff_t>((size_t)m + 2);
  uint32_t *nzrow = A1.get<uint32_t>((size_t)m + 2), *chunk_k = A1.get<uint32_t>((size_t)nchunks + 2);
  uint32_t *rowstart = A1.get<uint32_t>((size_t)(n >> 5) + 4);
  uint32_t *S = A1.get<uint32_t>(n);
  PT_CHECK_PTR(S);
  GDN_TRY(pt_zero(hist, (PB_HUB_BUCKETS + PB_LIN_BINS) * 4));
  GDN_TRY(pt_zero(rowstart, ((size_t)(n >> 5) + 4) * 4));
  hipLaunchKernelGGL(po_degrees_kernel, dim3(gdn_nblocks(m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, deg, cnt16);
  if (max_tiers > 0) {
    hipLaunchKernelGGL(pb_hub_hist_kernel, dim3(2048), dim3(GDN_BLOCK), 0, 0, cnt16, (size_t)m, hist);
    hipLaunchKernelGGL(pb_hub_hist_lin_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, cnt16, (size_t)m, hist + PB_HUB_BUCKETS);
  }
  hipLaunchKernelGGL(pt_row_count_kernel, dim3(nbV), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, bcR);
  hipLaunchKernelGGL(pt_scan_rows_kernel, dim3(1), dim3(GDN_BLOCK), 0, 0, bcR, nbV, totals + 8);
  GDN_HIP(hipGetLastError());
  PtSrcArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.src_count = deg;
  sa.cnt16 = cnt16;
  sa.n = m;
  unsigned h_tot[16] = {};
  if (max_tiers > 0) {
    std::vector<unsigned> h_hist(PB_HUB_BUCKETS + PB_LIN_BINS);
    GDN_HIP(hipMemcpy(h_hist.data(), hist, h_hist.size() * 4, hipMemcpyDeviceToHost));
    sa.ntiers = po_choose_tiers(h_hist.data(), h_hist.data() + PB_HUB_BUCKETS, a.tier_min_deg < 1 ? 1u : a.tier_min_deg, max_tiers, a.caps, sa.thr);
  }
  GDN_HIP(hipMemcpy(h_tot + 8, totals + 8, 4, hipMemcpyDeviceToHost));
  const eoff_t n_nz = h_tot[8];
  hipLaunchKernelGGL(pt_src_count_kernel, dim3(nbV), dim3(GDN_BLOCK), 0, 0, sa, nbV, bcS);
  hipLaunchKernelGGL(pt_scan_rows_kernel, dim3(PT_NCLS), dim3(GDN_BLOCK), 0, 0, bcS, nbV, totals);
  GDN_HIP(hipGetLastError());
  GDN_HIP(hipMemcpy(h_tot, totals, PT_NCLS * 4, hipMemcpyDeviceToHost));
  while (sa.ntiers > 0 && h_tot[sa.ntiers] == 0) sa.ntiers--;
  const unsigned ntiers = (unsigned)sa.ntiers;
  unsigned n_ts = 0;
  for (unsigned t = 0; t < ntiers; t++) {
    if (h_tot[1 + t] > a.caps[t] || h_tot[1 + t] == 0) {
      gdn_set_error("out layout: tier %u holds %u sources (cap %u)", t, h_tot[1 + t], a.caps[t]);
      return GDN_ERR_INVALID;
    }
    ts.off[t] = n_ts;
    n_ts += h_tot[1 + t];
  }
  ts.off[ntiers] = n_ts;
  phase("degrees, classes");
  const unsigned ncls = 1 + ntiers, nd = d1 + ntiers * dt;
  const unsigned long long ntiles = (unsigned long long)nchunks * nbins;
  p.m_local = g->m;
  p.m_global = g->m;
  p.log_chunk = lc;
  p.log_bin = lb;
  p.compact = false;
  p.chunk_slots = 1u << lc;
  p.log_group = a.log_group;
  p.nchunks = nchunks;
  p.nbins = nbins;
  GDN_TRY(p.chunk_ptr.alloc((size_t)nchunks + 1));
  GDN_TRY(p.bin_ptr.alloc((size_t)nbins + 1));
  GDN_TRY(p.errflag.alloc(1));
  GDN_HIP(hipMemsetAsync(p.errflag.p, 0, sizeof(unsigned), 0));
  if (n_ts) GDN_TRY(ts.ids.alloc(n_ts));
  if (ntiers) GDN_TRY(ts.ptr.alloc((size_t)ntiers * nbins + 1));
  PtArena A2;
  {
    size_t bytes = pt_pad256((size_t)ncls * ntiles * 4) + pt_pad256(ntiles * 4) * 2 + pt_pad256((ntiles + 1) * 8) * 2;
    bytes += pt_pad256((size_t)nchunks * nd * 4) + pt_pad256(((size_t)nchunks * nd + 1) * 8) + pt_pad256((size_t)(ntiers + 1) * nbins * 4);
    bytes += pt_pad256(((size_t)nchunks + 2) * 8) * 2 + pt_pad256(((size_t)nbins + 2) * 8) + pt_pad256((size_t)nchunks * 4);
    bytes += pt_pad256(((ntiles > (unsigned long long)nchunks * nd ? ntiles : (unsigned long long)nchunks * nd) / 2048 + 16) * 8);
    GDN_TRY(A2.init(bytes, 2));
  }
  uint32_t *cnt = A2.get<uint32_t>((size_t)ncls * ntiles), *psz_c = A2.get<uint32_t>(ntiles), *psz_b = A2.get<uint32_t>(ntiles);
  eoff_t *pu = A2.get<eoff_t>(ntiles + 1), *pv = A2.get<eoff_t>(ntiles + 1);
  uint32_t *segsz = A2.get<uint32_t>((size_t)nchunks * nd);
  eoff_t *segoff = A2.get<eoff_t>((size_t)nchunks * nd + 1);
  uint32_t *tsz = A2.get<uint32_t>((size_t)(ntiers + 1) * nbins);
  eoff_t *d_du = A2.get<eoff_t>((size_t)nchunks + 2), *chunk_sz = A2.get<eoff_t>((size_t)nchunks + 2), *d_dv = A2.get<eoff_t>((size_t)nbins + 2);
  uint32_t *order = A2.get<uint32_t>(nchunks);
  eoff_t *ws = A2.get<eoff_t>((ntiles > (unsigned long long)nchunks * nd ? ntiles : (unsigned long long)nchunks * nd) / 2048 + 16);
  PT_CHECK_PTR(ws);
  {
    PtSrcOut so;
    memset(&so, 0, sizeof(so));
    for (unsigned t = 0; t < ntiers; t++) so.ids[t] = ts.ids.p + ts.off[t];
    hipLaunchKernelGGL(po_src_assign_kernel, dim3(nbV), dim3(GDN_BLOCK), 0, 0, sa, nbV, bcS, smap, so);
    hipLaunchKernelGGL(po_row_assign_kernel, dim3(nbV), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, bcR, lc, crp, nzrow, rowstart, chunk_k);
    GDN_HIP(hipGetLastError());
  }
  {
    PoCountArgs ca;
    ca.rowptr = g->rowptr;
    ca.colidx = g->colidx;
    ca.smap = smap;
    ca.rowstart = reinterpret_cast<const unsigned long long *>(rowstart);
    ca.crp = crp;
    ca.nzrow = nzrow;
    ca.chunk_k = chunk_k;
    ca.n_nz = n_nz;
    ca.m = m;
    ca.nchunks = nchunks;
    ca.nbins = nbins;
    ca.ncls = ncls;
    ca.log_chunk = lc;
    ca.log_bin = lb;
    ca.S = S;
    ca.cnt = cnt;
    const size_t lds = (size_t)ncls * nbins * 4;
    GDN_HIP(hipFuncSetAttribute((const void *)po_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds > 65536 ? lds : 65536)));
    hipLaunchKernelGGL(po_count_kernel, dim3(nchunks), dim3(PT_THREADS), lds, 0, ca);
    GDN_HIP(hipGetLastError());
  }
  phase("po_count");
  eoff_t xlen = 0, n_te = 0, n_te_pad = 0;
  if (ntiers) {
    hipLaunchKernelGGL(po_tier_prefix_kernel, dim3(gdn_nblocks((uint64_t)ntiers * nbins)), dim3(GDN_BLOCK), 0, 0, cnt, nchunks, nbins, ntiers, tsz);
    GDN_TRY(gdn_exclusive_scan_u32_to_u64_ws(tsz, ts.ptr.p, (size_t)ntiers * nbins, ws, 0));
    GDN_HIP(hipMemcpy(&n_te, ts.ptr.p + (size_t)ntiers * nbins, sizeof(eoff_t), hipMemcpyDeviceToHost));
    ts.interleaved = a.interleave;
    if (a.interleave) {  // streams start on multiples of 256 records; the counts travel with the plan (no pad records)
      const unsigned nst = ntiers * nbins;
      DevBuf<uint32_t> tszp;
      GDN_TRY(ts.cnt.alloc(nst));
      GDN_TRY(tszp.alloc(nst));
      GDN_HIP(hipMemcpyAsync(ts.cnt.p, tsz, (size_t)nst * 4, hipMemcpyDeviceToDevice, 0));
      hipLaunchKernelGGL(po_pad_sizes_kernel, dim3(gdn_nblocks(nst)), dim3(GDN_BLOCK), 0, 0, tsz, nst, tszp.p);
      GDN_TRY(gdn_exclusive_scan_u32_to_u64_ws(tszp.p, ts.ptr.p, (size_t)nst, ws, 0));
      GDN_HIP(hipMemcpy(&n_te_pad, ts.ptr.p + (size_t)nst, sizeof(eoff_t), hipMemcpyDeviceToHost));
    } else n_te_pad = n_te;
  }
  // tiles padded so that a tile's candidates are whole 128-byte lines where tiles are long (gdn_sssp.hip)
  unsigned pad = a.pad;
  if (pad == 0) {
    const double avg_tile = (double)(n - n_te) / (double)ntiles;
    pad = avg_tile >= 1024.0 ? 128u : avg_tile >= 96.0 ? 64u : 32u;
  }
  GDN_REQUIRE(pad >= grp && pad <= 128 && (pad & (pad - 1)) == 0, "out layout: pad");
  {
    const unsigned long long tt = ntiles > (unsigned long long)nchunks * nd ? ntiles : (unsigned long long)nchunks * nd;
    // (po_sizes reads the per-chunk tier counts: po_tier_prefix has turned them into prefixes -- the segment sizes of the
    // tiers are taken from differences)
    hipLaunchKernelGGL(po_sizes_kernel, dim3(gdn_nblocks(tt)), dim3(GDN_BLOCK), 0, 0, cnt, nchunks, nbins, ntiers, pad, d1, dt, psz_c, psz_b, segsz, tsz);
    GDN_TRY(gdn_exclusive_scan_u32_to_u64_ws(psz_c, pu, (size_t)ntiles, ws, 0));
    GDN_TRY(gdn_exclusive_scan_u32_to_u64_ws(psz_b, pv, (size_t)ntiles, ws, 0));
    GDN_TRY(gdn_exclusive_scan_u32_to_u64_ws(segsz, segoff, (size_t)nchunks * nd, ws, 0));
    hipLaunchKernelGGL(pb_ptrs_kernel, dim3(gdn_nblocks((uint64_t)(nchunks > nbins ? nchunks : nbins) + 1)), dim3(GDN_BLOCK), 0, 0, pu, pv,
                       nchunks, nbins, p.chunk_ptr.p, p.bin_ptr.p);
    GDN_HIP(hipGetLastError());
  }
  std::vector<eoff_t> cs((size_t)nchunks + 1), bs((size_t)nbins + 1);
  GDN_HIP(hipMemcpy(cs.data(), p.chunk_ptr.p, cs.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
  GDN_HIP(hipMemcpy(bs.data(), p.bin_ptr.p, bs.size() * sizeof(eoff_t), hipMemcpyDeviceToHost));
  GDN_HIP(hipMemcpy(&xlen, segoff + (size_t)nchunks * nd, sizeof(eoff_t), hipMemcpyDeviceToHost));
  p.nnz = n - n_te;
  ts.edges = n_te;
  phase("offsets + readback");
  eoff_t n_pad = 0;
  std::vector<eoff_t> ca_((size_t)nchunks + 1, 0), ba((size_t)nbins + 1, 0);
  {
    std::vector<eoff_t> du(nchunks), dv(nbins);
    auto pick_align = [](eoff_t total, unsigned parts) {
      eoff_t al = 16;
      while (al < 16384 && al * 32 <= total / (parts ? parts : 1)) al <<= 1;
      return al;
    };
    const eoff_t al_c = pick_align(cs[nchunks], nchunks), al_b = pick_align(bs[nbins], nbins);
    for (unsigned c = 0; c < nchunks; c++) {
      du[c] = ca_[c] - cs[c];
      ca_[c + 1] = (ca_[c] + (cs[c + 1] - cs[c]) + al_c - 1) & ~(al_c - 1);
    }
    for (unsigned b = 0; b < nbins; b++) {
      dv[b] = ba[b] - bs[b];
      ba[b + 1] = (ba[b] + (bs[b + 1] - bs[b]) + al_b - 1) & ~(al_b - 1);
    }
    n_pad = ca_[nchunks] > ba[nbins] ? ca_[nchunks] : ba[nbins];
    GDN_HIP(hipMemcpyAsync(d_du, du.data(), du.size() * sizeof(eoff_t), hipMemcpyHostToDevice, 0));
    GDN_HIP(hipMemcpyAsync(d_dv, dv.data(), dv.size() * sizeof(eoff_t), hipMemcpyHostToDevice, 0));
    hipLaunchKernelGGL(pb_shift_kernel, dim3(gdn_nblocks(ntiles)), dim3(GDN_BLOCK), 0, 0, pu, pv, d_du, d_dv, nchunks, nbins);
    GDN_HIP(hipMemcpyAsync(p.chunk_ptr.p, ca_.data(), ca_.size() * sizeof(eoff_t), hipMemcpyHostToDevice, 0));
    GDN_HIP(hipMemcpyAsync(p.bin_ptr.p, ba.data(), ba.size() * sizeof(eoff_t), hipMemcpyHostToDevice, 0));
    // chunks by descending edge count (the launch order of the split): from the row offsets of the chunk starts
    std::vector<uint32_t> co(nchunks);
    for (unsigned c = 0; c < nchunks; c++) co[c] = c;
    std::stable_sort(co.begin(), co.end(), [&](uint32_t x, uint32_t y) { return cs[x + 1] - cs[x] > cs[y + 1] - cs[y]; });
    GDN_HIP(hipMemcpyAsync(order, co.data(), (size_t)nchunks * 4, hipMemcpyHostToDevice, 0));
    GDN_HIP(hipStreamSynchronize(0));
  }
  (void)chunk_sz;
  p.n_pad = n_pad;
  if ((n_pad >> a.log_group) + 1 > 0xFFFFFFFFull) {
    gdn_set_error("out layout: more than 2^35 padded edges");
    return GDN_ERR_INVALID;
  }
  PtArena A3;
  GDN_TRY(A3.init(pt_pad256((size_t)xlen * 4) * 2 + 4096, 4));
  uint32_t *X = A3.get<uint32_t>(xlen), *XV = A3.get<uint32_t>(xlen);
  PT_CHECK_PTR(XV);
  GDN_TRY(p.U.alloc(n_pad + grp));
  GDN_TRY(p.V.alloc(n_pad + grp));
  GDN_TRY(p.G.alloc((n_pad >> a.log_group) + 1));
  GDN_TRY(Wp.alloc(n_pad + grp));
  if (ntiers) {
    GDN_TRY(ts.rec.alloc((size_t)n_te_pad + 16));
    if (a.want_w8) GDN_TRY(ts.w8.alloc((size_t)n_te_pad + 16));
  }
  {
    const unsigned long long fb = (n_pad + grp + GDN_BLOCK - 1) / GDN_BLOCK;
    hipLaunchKernelGGL(pb_fill_u16_kernel, dim3((unsigned)(fb > 262144ull ? 262144ull : fb)), dim3(GDN_BLOCK), 0, 0, p.U.p, n_pad + grp,
                       (uint16_t)p.chunk_slots);
    GDN_HIP(hipMemsetAsync(p.V.p, 0, (n_pad + grp) * sizeof(uint16_t), 0));
    GDN_HIP(hipMemsetAsync(Wp.p, 0, (n_pad + grp) * sizeof(float), 0));
    const unsigned long long ng = (n_pad >> a.log_group) + 1;
    const unsigned long long fbg = (ng + GDN_BLOCK - 1) / GDN_BLOCK;
    hipLaunchKernelGGL(pb_fill_u32_kernel, dim3((unsigned)(fbg > 262144ull ? 262144ull : fbg)), dim3(GDN_BLOCK), 0, 0, p.G.p, ng,
                       (uint32_t)(n_pad >> a.log_group));
  }
  phase("arena 3 + arrays + fills");
  {
    PoSplitArgs sp;
    sp.rowptr = g->rowptr;
    sp.colidx = g->colidx;
    sp.S = S;
    sp.weight = a.weight;
    sp.segoff = segoff;
    sp.order = order;
    sp.X = X;
    sp.XV = XV;
    sp.m = m;
    sp.d1 = d1;
    sp.dt = dt;
    sp.ntiers = ntiers;
    sp.log_chunk = lc;
    sp.log_bin = lb;
    const size_t lds = po_stage_bytes(nd);
    GDN_HIP(hipFuncSetAttribute((const void *)po_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(po_split_kernel, dim3(nchunks), dim3(PT_PTHREADS), lds, 0, sp);
    GDN_HIP(hipGetLastError());
  }
  phase("po_split");
  {
    PoTilesArgs ta;
    ta.X = X;
    ta.XV = XV;
    ta.segoff = segoff;
    ta.pu = pu;
    ta.pv = pv;
    ta.tptr = ntiers ? ts.ptr.p : nullptr;
    ta.tpre = cnt;
    ta.U = p.U.p;
    ta.V = p.V.p;
    ta.W = reinterpret_cast<uint32_t *>(Wp.p);
    ta.rec = ntiers ? ts.rec.p : nullptr;
    ta.w8 = (ntiers && a.want_w8) ? ts.w8.p : nullptr;
    ta.nchunks = nchunks;
    ta.nbins = nbins;
    ta.d1 = d1;
    ta.dt = dt;
    ta.ntiers = ntiers;
    const unsigned long long nseg = (unsigned long long)nchunks * nd;
    hipLaunchKernelGGL(po_tiles_kernel, dim3(gdn_nblocks(nseg * 64)), dim3(GDN_BLOCK), 0, 0, ta);
    GDN_HIP(hipGetLastError());
  }
  if (ntiers && ts.interleaved) {
    hipLaunchKernelGGL(po_interleave_kernel, dim3(ntiers * nbins), dim3(GDN_BLOCK), 0, 0, ts.rec.p, a.want_w8 ? ts.w8.p : nullptr, ts.ptr.p, ts.cnt.p);
    GDN_HIP(hipGetLastError());
  }
  phase("po_tiles");
  hipLaunchKernelGGL(pb_groups_kernel, dim3(gdn_nblocks(ntiles)), dim3(GDN_BLOCK), 0, 0, pu, pv, psz_c, nchunks, nbins, p.G.p, 0, a.log_group);
  GDN_HIP(hipGetLastError());
  {
    std::vector<uint32_t> co(nchunks), bo(nbins);
    for (unsigned i = 0; i < nchunks; i++) co[i] = i;
    for (unsigned i = 0; i < nbins; i++) bo[i] = i;
    std::stable_sort(co.begin(), co.end(), [&](uint32_t x, uint32_t y) { return ca_[x + 1] - ca_[x] > ca_[y + 1] - ca_[y]; });
    std::stable_sort(bo.begin(), bo.end(), [&](uint32_t x, uint32_t y) { return ba[x + 1] - ba[x] > ba[y + 1] - ba[y]; });
    GDN_TRY(p.chunk_order.alloc(nchunks));
    GDN_TRY(p.bin_order.alloc(nbins));
    GDN_HIP(hipMemcpyAsync(p.chunk_order.p, co.data(), co.size() * 4, hipMemcpyHostToDevice, 0));
    GDN_HIP(hipMemcpyAsync(p.bin_order.p, bo.data(), bo.size() * 4, hipMemcpyHostToDevice, 0));
    GDN_HIP(hipStreamSynchronize(0));
  }
  GDN_TRY(p.partial.alloc(nbins));
  GDN_TRY(p.red_scratch.alloc(2 * ((size_t)nbins / 4096 + 2)));
  ts.n = (int)ntiers;
  GDN_HIP(hipDeviceSynchronize());
  phase("groups + orders");
  if (trace) {
    fprintf(stderr, "[pb_build_out] edges %llu: main %llu padded %llu chunks %u bins %u", n, (unsigned long long)p.nnz, (unsigned long long)n_pad, nchunks, nbins);
    for (unsigned t = 0; t < ntiers; t++) fprintf(stderr, ", tier %u: %u sources (degree >= %u)", t, h_tot[1 + t], sa.thr[t]);
    fprintf(stderr, ", %llu tier edges; %.2f ms wall\n", (unsigned long long)n_te,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  }
  return GDN_OK;
}
