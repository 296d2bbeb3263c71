// gdn_tc.hip -- triangle counting on the degree-oriented DAG.
//
// Reference path: TCSolver (src/tc/tc.h:7).  OpenMP src/tc/omp_base.cc:16-22 (for u, for v in
// N(u): |N(u) ^ N(v)| by merge, include/VertexSet.h:65-76); CUDA src/tc/gpu_base.cu:11
// warp_edge = one 32-lane warp per DAG edge, binary search of the shorter list in the longer
// (include/graph_gpu.h:253 warp_intersect_cache, include/search.cuh:45) + CUB BlockReduce.
// The orientation that the reference applies while loading (`Graph g(prefix, USE_DAG)`,
// src/tc/main.cc:12 -> src/common/graph.cc:67-113) is done on the device here:
//   keep[k] = deg[dst] > deg[src] || (deg equal && dst > src)        (graph.cc:80-81)
//   pos     = exclusive_scan(keep)                                    (order preserving)
//   dag.colidx[pos[k]] = colidx[k];  dag.rowptr[u] = pos[rowptr[u]]
// Counting: one wavefront per source vertex u with N+(u) staged in LDS; the neighbour lists of 64
// out-neighbours at a time are walked lane-packed (gdn_expand.hpp) and looked up by binary search in
// LDS; uint64 count reduced per wave then one atomicAdd per workgroup.  Exact (integer sum).
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "gdn_expand.hpp"

struct TcKeepVis {
  const vid_t *__restrict__ colidx;
  const int32_t *__restrict__ deg;
  unsigned *__restrict__ keep;
  int32_t v;  // per-lane source vertex
  __device__ __forceinline__ void begin_big(vid_t vv) { v = vv; }
  __device__ __forceinline__ void edge(int owner, eoff_t k, bool valid) {
    const int32_t src = __shfl(v, owner, 64);
    if (valid) {
      const vid_t dst = colidx[k];
      const int32_t ds = deg[src], dd = deg[dst];
      keep[k] = (dd > ds || (dd == ds && dst > src)) ? 1u : 0u;
    }
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
tc_keep_kernel(const eoff_t *__restrict__ rowptr, int32_t m, ExpBigList big, TcKeepVis vis) {
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
tc_keep_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, TcKeepVis vis) {
  vis.v = 0;
  gdn_expand_big_items(rowptr, big, vis);
}

__global__ void __launch_bounds__(GDN_BLOCK)
tc_compact_kernel(const vid_t *__restrict__ colidx, const unsigned *__restrict__ keep,
                  const eoff_t *__restrict__ pos, uint64_t nnz, vid_t *__restrict__ out) {
  size_t k = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * GDN_BLOCK;
  for (; k < nnz; k += stride)
    if (keep[k]) out[pos[k]] = colidx[k];
}

__global__ void __launch_bounds__(GDN_BLOCK)
tc_rowptr_kernel(const eoff_t *__restrict__ rowptr, const eoff_t *__restrict__ pos, int32_t m,
                 eoff_t *__restrict__ out) {
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v <= (unsigned)m) out[v] = pos[rowptr[v]];
}

// Counting.  One wavefront per source vertex u.  N+(u) is staged in LDS (TC_CAP ids per wave);
// the out-neighbours v of u are taken 64 at a time, one per lane, and THEIR neighbour lists are
// walked through gdn_expand_wave, so the 64 lanes always hold 64 distinct (v, w) pairs no matter how
// short the lists are; every w is looked up in N+(u) by binary search in LDS (global memory when
// N+(u) does not fit).  With the degree orientation every list is O(sqrt(nnz)) long.
#ifndef TC_CAP
#define TC_CAP 1024            // ids of N+(u) staged in LDS per pass and wave (8 KB hash set -> 4 workgroups per CU)
#endif
#define TC_HASH (2 * TC_CAP)   // open-addressing slots per wave (load factor <= 0.5)
#define TC_EMPTY (-1)
#define TC_NOKEY (-2)  // a key no set holds (ids are >= 0): what a position past its list's end looks up (TcSet::count_fast)

#define TC_HASH_BITS (TC_CAP == 1024 ? 11 : TC_CAP == 512 ? 10 : TC_CAP == 256 ? 9 : 8)
static_assert((1 << TC_HASH_BITS) == TC_HASH, "TC_CAP must be 128, 256, 512 or 1024");
__device__ __forceinline__ unsigned tc_hash(vid_t w) { return ((unsigned)w * 2654435761u) >> (32 - TC_HASH_BITS); }

// membership of w in the staged part of N+(u): LDS hash set of 4-slot BUCKETS (TC_HASH / 4 of them), a bucket is one
// aligned 16-byte LDS read.  Keys go into the first free slot of their bucket (slots fill in order, so "slot 3 taken" =
// "bucket full") and spill into the next bucket only then: at the usual loads (a few hundred ids in 2048 slots) 99.7 % of
// the lookups -- hits and misses -- are settled by ONE read and four compares.  With one slot per probe and linear
// probing every wave instruction had some lane that needed a second and third dependent LDS round trip inside a
// divergent loop (a slot is taken with probability = the load factor, and 64 lanes probe at once): those loops and their
// exec-mask bookkeeping were most of the 53 VALU + 43 SALU instructions per 64 probes (profiles/r02_tc_pmc.md).
typedef int tc_i32x4 __attribute__((ext_vector_type(4)));
#define TC_BUCKETS (TC_HASH / 4)
// (a 24-bit multiply -- v_mul_u32_u24 is full rate where v_mul_lo_u32 is quarter rate -- of the id folded onto itself was
// measured: 3 % SLOWER on RMAT-21 / 23, it spreads the ids less evenly over the buckets)
#ifdef TC_HASH_MUL24  // A/B: the full-rate 24-bit multiply (ids below 2^24)
__device__ __forceinline__ unsigned tc_bucket(vid_t w) { return __umul24((unsigned)w, 0x9E3779u) >> (32 - (TC_HASH_BITS - 2)); }
#elif defined(TC_HASH_XOR)  // A/B (tools/build_variant.sh): two full-rate instructions instead of the quarter-rate 32-bit multiply
__device__ __forceinline__ unsigned tc_bucket(vid_t w) { return (((unsigned)w >> 9) ^ (unsigned)w) & (TC_BUCKETS - 1u); }
#else
__device__ __forceinline__ unsigned tc_bucket(vid_t w) { return ((unsigned)w * 2654435761u) >> (32 - (TC_HASH_BITS - 2)); }
#endif

struct TcSet {
  const vid_t *table;  // 16-byte aligned
  // N candidates at once: the N bucket reads are independent (in flight together); the rare spill-over is followed for
  // all of them together, one bucket per wave-uniform round.  `vm[r]` = ballot of the lanes whose candidate r is real.
  // All the bookkeeping -- hit counts, pending sets -- is done on the 64-bit lane masks, i.e. on the scalar unit: the
  // count is a wave total anyway (the caller adds the return value ONCE per wave, not per lane), and flags kept as 0/1
  // values in vector registers were a third of the loop's VALU instructions.
  template <int N>
  __device__ __forceinline__ unsigned count(const vid_t (&w)[N], const unsigned long long (&vm)[N]) const {
    unsigned c = 0;
    unsigned hb[N];
    tc_i32x4 b[N];
    unsigned long long pend[N];
#pragma unroll
    for (int r = 0; r < N; r++) {
      hb[r] = tc_bucket(w[r]);
      b[r] = *reinterpret_cast<const tc_i32x4 *>(table + 4u * hb[r]);
    }
    unsigned long long any = 0ull;
#pragma unroll
    for (int r = 0; r < N; r++) {
      // four ballots of plain compares OR-ed on the scalar unit (a ballot of the OR-ed condition makes the compiler
      // materialise the mask in a vector register and compare it again: two more VALU instructions per chunk)
      const unsigned long long hit = (__ballot(b[r].x == w[r]) | __ballot(b[r].y == w[r]) | __ballot(b[r].z == w[r]) |
                                      __ballot(b[r].w == w[r])) & vm[r];
      c += (unsigned)__popcll(hit);
      pend[r] = __ballot(b[r].w != TC_EMPTY) & vm[r] & ~hit;  // bucket full and not found: look in the next one
      any |= pend[r];
    }
    while (any) {
      any = 0ull;
#pragma unroll
      for (int r = 0; r < N; r++) {
        if (pend[r]) {  // uniform
          hb[r] = (hb[r] + 1u) & (TC_BUCKETS - 1);
          b[r] = *reinterpret_cast<const tc_i32x4 *>(table + 4u * hb[r]);
          const unsigned long long hit = (__ballot(b[r].x == w[r]) | __ballot(b[r].y == w[r]) | __ballot(b[r].z == w[r]) |
                                          __ballot(b[r].w == w[r])) & pend[r];
          c += (unsigned)__popcll(hit);
          pend[r] = __ballot(b[r].w != TC_EMPTY) & pend[r] & ~hit;
          any |= pend[r];
        }
      }
    }
    return c;  // wave total (the same in every lane)
  }
  // The same look-ups WITHOUT lane masks for the list ends, for the chunk stream of the long lists (round 6).  With its work
  // counters out of the way (profiles/r06_tc_counters.md) the kernel turned out to be bound by neither bytes nor round trips --
  // every long list served from a 1 MB window: 10.6 instead of 10.8 ms -- but by instruction ISSUE at four waves per SIMD: 37
  // scalar and 30 vector instructions per chunk of 64 look-ups, and a SIMD issues one of each per four cycles (scalar 65 %,
  // vector 52 % of the kernel's cycles).  Here a position past its list's end carries a key no set holds (TC_NOKEY) instead of
  // a cleared mask bit -- the scalar iterator computes no masks -- and hits are counted per LANE, one add with the hit mask as
  // carry, instead of a popcount and an add on the scalar unit; the caller sums the lanes once per walk.  (All-vector forms --
  // min(slot ^ key) == 0, the pending test as an AND-reduction -- were measured first: 17 scalar but 37-45 vector instructions
  // per chunk, hash-set kernel alone 11.8 / 10.1 ms, the Orkut-like count 3.7 % slower: sessions r06_30, r06_32.)
  template <int N>
  __device__ __forceinline__ void count_fast(const vid_t (&w)[N], unsigned &cl) const {
    unsigned hb[N];
    tc_i32x4 b[N];
    unsigned long long pend[N];
#pragma unroll
    for (int r = 0; r < N; r++) {
      hb[r] = tc_bucket(w[r]);
      b[r] = *reinterpret_cast<const tc_i32x4 *>(table + 4u * hb[r]);
    }
    unsigned long long any = 0ull;
#pragma unroll
    for (int r = 0; r < N; r++) {
      const unsigned long long hit = __ballot(b[r].x == w[r]) | __ballot(b[r].y == w[r]) | __ballot(b[r].z == w[r]) |
                                     __ballot(b[r].w == w[r]);
      asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(cl) : "s"(hit) : "vcc");  // cl += this lane's bit of `hit`
      pend[r] = __ballot(b[r].w != TC_EMPTY) & ~hit;  // bucket full and not found: look in the next one
      any |= pend[r];
    }
    if (__builtin_expect(any != 0ull, 0)) {  // (one look-up in ~300 at the usual loads: every second step of 256)
#pragma unroll
      for (int r = 0; r < N; r++) {
        if (pend[r]) pend[r] &= __ballot(w[r] != TC_NOKEY);  // positions past a list's end (all in ONE bucket) are not followed
        while (pend[r]) {  // uniform
          hb[r] = (hb[r] + 1u) & (TC_BUCKETS - 1);
          b[r] = *reinterpret_cast<const tc_i32x4 *>(table + 4u * hb[r]);
          const unsigned long long hit = (__ballot(b[r].x == w[r]) | __ballot(b[r].y == w[r]) | __ballot(b[r].z == w[r]) |
                                          __ballot(b[r].w == w[r])) & pend[r];
          asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(cl) : "s"(hit) : "vcc");
          pend[r] = __ballot(b[r].w != TC_EMPTY) & pend[r] & ~hit;
        }
      }
    }
  }
};

// insert / remove one key (each lane its own; keys of one list are distinct)
__device__ __forceinline__ void tc_insert(vid_t *tab, vid_t x) {
  unsigned h = tc_bucket(x);
  for (;;) {
#pragma unroll
    for (int sl = 0; sl < 4; sl++)
      if (atomicCAS(&tab[4u * h + sl], TC_EMPTY, x) == TC_EMPTY) return;
    h = (h + 1u) & (TC_BUCKETS - 1);
  }
}
__device__ __forceinline__ void tc_remove(vid_t *tab, vid_t x) {
  unsigned h = tc_bucket(x);
  for (;;) {  // x is there until THIS lane removes it: a bucket that does not hold it was full when x arrived
#pragma unroll
    for (int sl = 0; sl < 4; sl++)
      if (tab[4u * h + sl] == x) {
        tab[4u * h + sl] = TC_EMPTY;
        return;
      }
    h = (h + 1u) & (TC_BUCKETS - 1);
  }
}

// neighbour-list elements per lane in flight: the kernel is latency bound (67 % of its wave cycles were spent in
// s_waitcnt with one element per lane, profiles/r01_tc_pmc.md)
#ifndef TC_UNR
#define TC_UNR 4  // chunks per step; two steps are in flight (measured on RMAT-21 / 23: 4 and 8 alike, 16 spills: 3x slower)
#endif
#ifndef TC_LONG
#define TC_LONG 48  // lists at least this long are walked in 64-element chunks, shorter ones packed
#endif
#ifndef TC_WAVES_PER_EU
// LDS allows 4 workgroups per CU (8 waves/SIMD with 4 KB sets measured slower).  The kernel's LDS arrays are DYNAMIC, so that the
// compiler does not see that cap, and the bound asked for is five waves per SIMD: the code object then allocates 93 vector
// registers (.vgpr_count, llvm-readelf --notes) where the build with static arrays -- the compiler takes what four waves allow --
// allocates 97.  Registers come in blocks of 8: four hash-set waves of 96 leave 128 of a SIMD's 512 to the core kernel, TWO of its
// waves of 56; four of 104 leave 96, ONE.  Beside the core kernel the count is 15 % faster on R-MAT graphs (RMAT-23 12.7 -> 10.8 ms,
// RMAT-24 31.8 -> 28.5; Orkut-like 1 % slower), each kernel ALONE takes the same time in both builds (round 6,
// profiles/r06_tc_counters.md section 7).  Bounds 4 / 5 / 6 with dynamic arrays: 93 / 93 / 80 registers, RMAT-23 10.80 / 10.77 /
// 10.57 ms, Orkut-like 6.79 / 6.79 / 7.50 (spills at 80), session r06_68.  KEEP THE KERNEL AT <= 96 REGISTERS.
#define TC_WAVES_PER_EU 5
#endif

// Walk the out-neighbour lists [vb,ve) held one per lane (vb == ve for idle lanes) and count the elements that
// are in `set`.  Lists of >= 64 elements are walked by the whole wave, shorter ones are packed (wave prefix sum of
// the lengths; the owner of a packed position comes from start markers dropped into LDS instead of the 6-step
// binary search of gdn_expand.hpp: the kernel was VALU bound on that search) with TC_UNR independent positions
// per lane, so TC_UNR global loads and LDS probes overlap.
__device__ __forceinline__ unsigned long long tc_walk_lists(const vid_t *__restrict__ colidx, eoff_t vb, unsigned deg,
                                                            const TcSet &set, unsigned char *s_own) {
  const unsigned lane = gdn_lane();
  unsigned long long count = 0;
  {  // Lists of TC_LONG elements or more, cut into 64-element CHUNKS; the chunks of all of them form one stream that is
     // walked TC_UNR chunks per step, with the NEXT step's loads issued before this step's probes (two steps in flight per
     // wave).  Walking one list at a time in 512-slot steps left 45-60 % of the slots empty (the lists of a degree-ordered
     // DAG are 100-1000 long) and made every step pay a full memory round trip on its own: the count was bound by that
     // latency at 32 KB in flight per CU (profiles/r02_tc_pmc.md).  A chunk's list is wave-uniform: one ballot finds it,
     // its base pointer is scalar, positions are 32-bit offsets, loads are unpredicated (a position past the end reads the
     // list's last element and is masked in `valid`).
    const unsigned nch = deg >= (unsigned)TC_LONG ? (deg + 63u) >> 6 : 0u;
    const unsigned total = gdn_wave_sum(nch);
    const unsigned total_s = (unsigned)__builtin_amdgcn_readfirstlane((int)total);  // scalar copy: uniform branches
    // The chunk stream as a SCALAR iterator (round 3): which list a chunk belongs to, its base pointer, its length and the
    // position reached are wave-uniform, so they live in scalar registers and advance with a handful of scalar
    // instructions per chunk -- the first form found every chunk's list again (ballot of "owns chunks and starts at or
    // before q", four readlanes, 64-bit address arithmetic per lane: ~12 VALU + ~15 SALU per chunk, more than the look-up
    // itself; profiles/r03_tc_pmc.md: VALU issue was half of the kernel).  Per chunk now: lane offset, clamp, shift, load.
    // Every call ISSUES its load (past the end: the last element again, empty lane mask), so that the compiler can count
    // the outstanding loads and wait for one step's only.
    unsigned long long rem = __ballot(nch > 0u);  // lists not started yet (scalar)
    const vid_t *s_base = colidx;                 // current list (scalar)
    unsigned s_len = 1u, s_off = 1u;              // its length and the position of the next chunk (s_off >= s_len: finished)
    auto next_chunk = [&](vid_t &w) {
      if (s_off >= s_len && rem) {  // uniform: the next list
        const int owner = __ffsll((long long)rem) - 1;
        rem &= rem - 1ull;
        const eoff_t ob = ((eoff_t)(unsigned)__builtin_amdgcn_readlane((int)(vb >> 32), owner) << 32) |
                          (unsigned)__builtin_amdgcn_readlane((int)vb, owner);
#if defined(TC_ABL) && TC_ABL == 8  // timing-only ablation: every long list read from one 1 MB window (what the walks cost when L2 serves them)
        s_base = colidx + (ob & 0x3FFFFull);
#else
        s_base = colidx + ob;
#endif
        s_len = (unsigned)__builtin_amdgcn_readlane((int)deg, owner);
        s_off = 0u;
      }
      const unsigned o = s_off + lane;
      // (the clamp as one v_min_u32 against the scalar instead of compare + select, and the bucket hash as a full-rate 24-bit
      // multiply, were measured on one box: + 0.7 % and + 9 % -- session r06_60; fewer instructions are not always fewer cycles here)
      const vid_t x = s_base[o < s_len ? o : s_len - 1u];
      w = o < s_len ? x : TC_NOKEY;  // (past the end: a key no set holds -- no lane masks in the look-ups, TcSet::count_fast)
      s_off = s_off < s_len ? s_off + 64u : s_off;
    };
    auto load_step = [&](vid_t (&w)[TC_UNR]) {
#pragma unroll
      for (int r = 0; r < TC_UNR; r++) next_chunk(w[r]);
    };
    if (total_s) {
      vid_t w0[TC_UNR], w1[TC_UNR];
      unsigned cl = 0;  // hits of this lane
      load_step(w0);
      for (unsigned q0 = 0; q0 < total_s; q0 += 2 * TC_UNR) {
        load_step(w1);
        set.count_fast(w0, cl);
        load_step(w0);
        set.count_fast(w1, cl);
      }
      count += cl;  // (per lane: the caller's block sum takes every lane's count)
    }
    if (nch) deg = 0;
  }
  {  // shorter lists, packed
#if defined(TC_ABL) && TC_ABL == 6  // timing-only ablation: no packed path
    deg = 0;
#endif
    const unsigned incl = gdn_wave_incl_scan(deg);
    const unsigned total = __shfl(incl, 63, 64);
    if (total) {
      const unsigned excl = incl - deg;
      int carry = 0;  // owner of the last packed position handled so far (uniform)
      for (unsigned base = 0; base < total; base += 64 * TC_UNR) {
        unsigned idx[TC_UNR];
        int own[TC_UNR];
        bool valid[TC_UNR];
        // owner of a packed position WITHOUT a search: every non-empty list drops its lane id at the position
        // where it starts (positions are distinct), a position's owner is the nearest marker at or before it
        const unsigned mypos = excl - base;  // wraps for lists that started before this step
        const bool starts_here = deg > 0u && mypos < 64u * TC_UNR;
        if (starts_here) s_own[mypos] = (unsigned char)(lane + 1u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned mark[TC_UNR];
#pragma unroll
        for (int r = 0; r < TC_UNR; r++) mark[r] = s_own[64 * r + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (starts_here) s_own[mypos] = 0;
#pragma unroll
        for (int r = 0; r < TC_UNR; r++) {
          idx[r] = base + 64u * (unsigned)r + lane;
          valid[r] = idx[r] < total;
          const unsigned long long starts = __ballot(mark[r] != 0u);
          const unsigned long long upto = starts & (gdn_lanemask_lt() | (1ull << lane));
          const int from = 63 - __clzll((long long)upto);  // lane of the nearest marker (garbage when upto == 0)
          const int fetched = (int)__shfl(mark[r], from & 63, 64) - 1;
          own[r] = upto ? fetched : carry;
          carry = __shfl(own[r], 63, 64);
        }
        vid_t w[TC_UNR];
#pragma unroll
        for (int r = 0; r < TC_UNR; r++) {
          const int owner = valid[r] ? own[r] : (int)lane;
          const eoff_t ob = __shfl(vb, owner, 64);
          const unsigned oex = __shfl(excl, owner, 64);
          w[r] = valid[r] ? colidx[ob + (idx[r] - oex)] : 0;
        }
        unsigned long long vmask[TC_UNR];
#pragma unroll
        for (int r = 0; r < TC_UNR; r++) vmask[r] = __ballot(valid[r]);
        const unsigned cw = set.count(w, vmask);
        if (lane == 0) count += cw;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  return count;
}

// One wave: count the triangles closed by the out-neighbours v = colidx[vlo..vhi) of the row [ub,ue) (vlo..vhi is
// the whole row for a light row, a slice of it for a heavy one).  N+(u) is staged in the wave's LDS hash set TC_CAP
// ids at a time (one pass for all but a few hub rows; every pass walks all the lists of the slice).
// ncol: the array the neighbours [vlo,vhi) are read from -- colidx itself (the row's own list: "for u, for v in N+(u)") or
// the in-CSR's column ids (the v-centric count: the set is N+(v), the neighbours are the u with u -> v, see tc_count_kernel).
// nstart (nullable, parallel to ncol): neighbour i's list is walked from its element nstart[i] on (the FORWARD form: ids are
// degree ranks, the set is N+(v), and the elements of N+(u) up to and including v itself cannot be in it)
__device__ __forceinline__ unsigned long long tc_row_slice(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx,
                                                           const vid_t *__restrict__ ncol, eoff_t ub, eoff_t ue, eoff_t vlo,
                                                           eoff_t vhi, vid_t *s_tab, unsigned char *s_own,
                                                           const unsigned *__restrict__ nstart = nullptr,
                                                           const unsigned long long *__restrict__ nbound = nullptr) {
  const unsigned lane = gdn_lane();
  const int du = (int)(ue - ub);
  unsigned long long count = 0;
  TcSet set;
  set.table = s_tab;
  for (int c0 = 0; c0 < du; c0 += TC_CAP) {
    const int cn = du - c0 < TC_CAP ? du - c0 : TC_CAP;
    const eoff_t cb = ub + (eoff_t)c0;
    vid_t x0 = 0;  // the first 64 ids of the pass: loaded once for the hash build and the clean-up
    if (lane < (unsigned)cn) x0 = colidx[cb + lane];
    // the first 64 neighbours and their list bounds, requested BEFORE the set is built: behind the build's fence they were a
    // third and fourth dependent round trip of every row (set ids -> neighbour ids -> bounds -> lists), and a light row is
    // little more than that chain
    eoff_t vb0 = 0;    // (a walk as its first element and its length: a register less per walk than two offsets, and the kernel
    unsigned vd0 = 0;  // is held to 96 of them, see tc_count_rows)
    if (nbound) {  // the forward form with packed walk bounds: no neighbour id, no row offsets
      if (vlo + lane < vhi) {
        const unsigned long long b = nbound[vlo + lane];
        vb0 = b >> 24;
        vd0 = (unsigned)(b & 0xFFFFFFull);
      }
    } else if (vlo + lane < vhi) {
      const vid_t v = (ncol == colidx && vlo == cb) ? x0 : ncol[vlo + lane];
      vb0 = rowptr[v];
      const eoff_t e0 = rowptr[v + 1];
      if (nstart) vb0 += nstart[vlo + lane];
      vd0 = (unsigned)(e0 - vb0);
    }
    for (int i = lane; i < cn; i += 64) {  // build: integer LDS CAS, linear probing
      tc_insert(s_tab, i < 64 ? x0 : colidx[cb + i]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if defined(TC_ABL) && TC_ABL == 5  // timing-only ablation: set build + clear only, no walk
    for (eoff_t i0 = vlo; i0 < vlo; i0 += 64) {
#else
    for (eoff_t i0 = vlo; i0 < vhi; i0 += 64) {
#endif
      const eoff_t i = i0 + lane;
      eoff_t vb = vb0;
      unsigned vd = vd0;
      if (i0 != vlo) {
        vb = 0;
        vd = 0;
        if (i < vhi && nbound) {
          const unsigned long long b = nbound[i];
          vb = b >> 24;
          vd = (unsigned)(b & 0xFFFFFFull);
        } else if (i < vhi) {
          const vid_t v = (ncol == colidx && i0 == cb) ? x0 : ncol[i];
          vb = rowptr[v];
          const eoff_t e1 = rowptr[v + 1];
          if (nstart) vb += nstart[i];
          vd = (unsigned)(e1 - vb);
        }
      }
#if defined(TC_ABL) && TC_ABL == 7  // timing-only ablation: neighbour ids + bounds loaded, lists not walked
      count += (unsigned long long)(vd & 1u);
#else
      count += tc_walk_lists(colidx, vb, vd, set, s_own);
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < cn; i += 64) {  // clear only the slots that were used
      tc_remove(s_tab, i < 64 ? x0 : colidx[cb + i]);
    }
    // (a removal in progress leaves holes, but every key is still present until ITS lane removes it, and the search
    // scans whole buckets: all keys are found and the table is empty again afterwards.  Holes never outlive the pass.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  return count;
}

// Rows with more than `light` out-neighbours are cut into work items of TC_SLICE neighbours: the work of a row
// grows with du * (length of its neighbours' lists), and on a skewed graph a third of all list elements belongs to
// a few thousand hub rows -- one wave per row left the kernel waiting for them (profiles/r01_tc_pmc.md).
// (the limit is a launch argument: measured on symmetrized R-MAT, count kernel only, limit 64 / 256 / 512: scale 19
// 2.00 / 1.85 / 2.17 ms, scale 21 10.4 / 10.08 / 9.76, scale 23 68.8 / 66.3 / 66.3 -- fewer set rebuilds and item grabs
// with a higher limit, until whole rows on single waves make the tail of a small graph)
// Work items are handed out by atomic counters, and a counter all 4096 resident waves share is a queue: device-scope atomics on
// ONE address retire at ~100 M/s (they are executed behind the XCDs' L2s), so the 524 K batch grabs + the item grabs of an
// RMAT-23 count were ~8 ms of atomics -- the kernel with its walks ablated away still took 8.1 of its 15.5 ms
// (profiles/r06_tc_ablation.md).  TC_NCUR counters, 128 bytes apart: unit B belongs to counter B mod TC_NCUR (its k-th unit is
// B = k TC_NCUR + c), a wave starts at the counter its number selects and moves on when one runs dry; every counter is seen dry
// at most once by a wave, after TC_NCUR of those everything is handed out.  Units keep their order inside a counter (rows are
// degree-ranked: the long ones come last on every counter alike).
#ifndef TC_NCUR
#define TC_NCUR 64
#endif
#define TC_CUR_STRIDE 32  // words between two counters
#ifndef TC_BATCH
#define TC_BATCH 16  // light rows per grab (<= 63: the batch's row offsets are one load)
#endif
struct TcGrab {
  unsigned *cur;
  unsigned c, dry;
  __device__ __forceinline__ TcGrab(unsigned *cursors, unsigned wave) : cur(cursors), c(wave % TC_NCUR), dry(0) {}
  // the next unit of [0, n), ~0u when all are handed out (wave-uniform)
  __device__ __forceinline__ unsigned next(unsigned n, unsigned lane) {
    while (dry < TC_NCUR) {
      unsigned k = 0;
      if (lane == 0) k = atomicAdd(&cur[c * TC_CUR_STRIDE], 1u);
      k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
      const unsigned long long unit = (unsigned long long)k * TC_NCUR + c;
      if (unit < n) return (unsigned)unit;
      c = (c + 1u) % TC_NCUR;
      dry++;
    }
    return ~0u;
  }
};

#define TC_LIGHT_MIN 64
#ifndef TC_SLICE
#define TC_SLICE 512  // (256 -> 512: 2 % faster, fewer set rebuilds; 1024 the same)
#endif

__global__ void __launch_bounds__(GDN_BLOCK)
tc_heavy_items_kernel(const eoff_t *__restrict__ rowptr, const eoff_t *__restrict__ nrowptr, int32_t row_lo, int32_t row_hi,
                      unsigned long long *__restrict__ items, unsigned capacity, unsigned *__restrict__ n_items,
                      unsigned *__restrict__ overflow, unsigned light) {
  const unsigned u = (unsigned)row_lo + blockIdx.x * GDN_BLOCK + threadIdx.x;  // rows [row_lo, row_hi)
  eoff_t du = 0;  // neighbours of the row (what the items slice); a row without a set closes nothing
  if (u < (unsigned)row_hi && rowptr[u + 1] > rowptr[u]) du = nrowptr[u + 1] - nrowptr[u];
  const unsigned n = du > light ? (unsigned)((du + TC_SLICE - 1) / TC_SLICE) : 0u;
  // wave-aggregated reservation: one atomic per wave
  const unsigned incl = gdn_wave_incl_scan(n);
  const unsigned tot = __shfl(incl, 63, 64);
  if (tot == 0) return;
  unsigned base = 0;
  if (gdn_lane() == 63) base = atomicAdd(n_items, tot);
  base = __shfl(base, 63, 64) + incl - n;
  for (unsigned c = 0; c < n; c++) {
    if (base + c < capacity) items[base + c] = ((unsigned long long)c << 32) | u;
    else *overflow = 1u;
  }
}

#ifdef TC_NUM_VGPR  // A/B (tools/build_variant.sh): the hash-set kernel held to this many vector registers (what is left of a SIMD's 512 is the core kernel's)
__attribute__((amdgpu_num_vgpr(TC_NUM_VGPR)))
#endif
__global__ void __launch_bounds__(GDN_BLOCK, TC_WAVES_PER_EU)
// Two formulations, same total.  u-centric (nrowptr == rowptr, ncolidx == colidx; the reference's loop, src/tc/omp_base.cc:
// 16-22): row u, set N+(u), neighbours v in N+(u), the elements of N+(v) are looked up -- SUM over edges of d+(v) probes.
// v-centric (nrowptr / ncolidx = the TRANSPOSED DAG): row v, set N+(v), neighbours the u with u -> v, the elements of
// N+(u) are looked up -- SUM over edges of d+(u) probes.  With the degree orientation v is the endpoint of higher degree
// and its out-list the longer one (R-MAT-19: 1.71 G against 1.01 G probes): gdn_tc_dev counts both sums and transposes
// the DAG when the v-centric form is the cheaper one.
tc_count_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const eoff_t *__restrict__ nrowptr,
                const vid_t *__restrict__ ncolidx, int32_t m,
                const unsigned long long *__restrict__ items, const unsigned *__restrict__ n_items_p,
                unsigned *__restrict__ cursors /* TC_NCUR heavy-item cursors, then TC_NCUR light-batch cursors (tc_grab) */,
                unsigned long long *__restrict__ total, unsigned light, const unsigned *__restrict__ nstart = nullptr,
                const unsigned long long *__restrict__ nbound = nullptr, int32_t row_lo = 0) {
  // (dynamic LDS: see TC_WAVES_PER_EU)  per wave: the hash set, the start markers of the packed lists (0 = none); a reduction word
  extern __shared__ __attribute__((aligned(16))) unsigned char tc_dyn[];
  vid_t(*s_tab)[TC_HASH] = reinterpret_cast<vid_t(*)[TC_HASH]>(tc_dyn);
  unsigned long long *s_red = reinterpret_cast<unsigned long long *>(tc_dyn + sizeof(vid_t) * GDN_WAVES_PER_BLOCK * TC_HASH);
  unsigned char(*s_own)[64 * TC_UNR] =
      reinterpret_cast<unsigned char(*)[64 * TC_UNR]>(tc_dyn + sizeof(vid_t) * GDN_WAVES_PER_BLOCK * TC_HASH + 8 * GDN_WAVES_PER_BLOCK);
  const unsigned lane = gdn_lane();
  const unsigned w = threadIdx.x >> 6;
  unsigned long long count = 0;
  for (int i = lane; i < TC_HASH; i += 64) s_tab[w][i] = TC_EMPTY;
  for (int i = lane; i < 64 * TC_UNR; i += 64) s_own[w][i] = 0;
  // ---- heavy rows first (they are the long work items): one (row, slice) item per grab
#if defined(TC_ABL) && TC_ABL == 4  // timing-only ablation: no heavy items
  const unsigned n_items = 0;
#else
  const unsigned n_items = *n_items_p;
#endif
  const unsigned gw = blockIdx.x * GDN_WAVES_PER_BLOCK + w;
  TcGrab hg(cursors, gw);
  for (;;) {
    const unsigned it = hg.next(n_items, lane);
    if (it == ~0u) break;
    const unsigned long long item = items[it];
    const unsigned u = (unsigned)(item & 0xFFFFFFFFull), c = (unsigned)(item >> 32);
    const eoff_t ub = rowptr[u], ue = rowptr[u + 1];
    const eoff_t nb0 = nrowptr[u], ne0 = nrowptr[u + 1];
    const eoff_t vlo = nb0 + (eoff_t)c * TC_SLICE;
    const eoff_t vhi = vlo + TC_SLICE < ne0 ? vlo + TC_SLICE : ne0;
    count += tc_row_slice(rowptr, colidx, ncolidx, ub, ue, vlo, vhi, s_tab[w], s_own[w], nstart, nbound);
  }
  // ---- light rows: 16 consecutive vertices per grab (one atomic per 16 rows)
  const unsigned n_batches = ((unsigned)(m - row_lo) + TC_BATCH - 1u) / TC_BATCH;
  TcGrab lg(cursors + TC_NCUR * TC_CUR_STRIDE, gw);
  for (;;) {
#if defined(TC_ABL) && TC_ABL == 3  // timing-only ablation: no light rows
    break;
#endif
    const unsigned bt = lg.next(n_batches, lane);
    if (bt == ~0u) break;
    const unsigned u0 = (unsigned)row_lo + bt * TC_BATCH;
    const unsigned u1 = u0 + TC_BATCH < (unsigned)m ? u0 + TC_BATCH : (unsigned)m;
    eoff_t rp = 0, np = 0;  // the 17 row offsets of the batch in one load (per array)
    if (u0 + lane <= u1) {
      rp = rowptr[u0 + lane];
      np = nrowptr == rowptr ? rp : nrowptr[u0 + lane];
    }
    for (unsigned u = u0; u < u1; u++) {
      // (the row's four offsets are wave-uniform: read into scalar registers, not shuffled into vector ones)
      auto lane_u64 = [](eoff_t x, unsigned l) -> eoff_t {
        return ((eoff_t)(unsigned)__builtin_amdgcn_readlane((int)(x >> 32), (int)l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)x, (int)l);
      };
      const eoff_t ub = lane_u64(rp, u - u0), ue = lane_u64(rp, u - u0 + 1u);
      const eoff_t nb0 = lane_u64(np, u - u0), ne0 = lane_u64(np, u - u0 + 1u);
      const eoff_t dn = ne0 - nb0;
      // nothing to close without a set or a neighbour; heavy rows are done
      if (ue == ub || dn == 0 || dn > light) continue;
      if (nrowptr == rowptr && dn < 2) continue;  // u-centric: a single out-neighbour closes no triangle
      count += tc_row_slice(rowptr, colidx, ncolidx, ub, ue, nb0, ne0, s_tab[w], s_own[w], nstart, nbound);
    }
  }
  count = gdn_block_sum(count, s_red);
  if (threadIdx.x == 0 && count) atomicAdd(total, count);
}

// ------------------------------------------------------------------------------------------
// The north star's formulation, kept as a measured alternative (GDN_TC_FORM=bs): one WAVEFRONT per DAG edge (u, v),
// the elements of the SHORTER of N+(u) / N+(v) are looked up in the LONGER by binary search -- the reference's
// warp_edge kernel (src/tc/gpu_base.cu:11-23 over include/graph_gpu.h:253 warp_intersect_cache and
// include/search.cuh:45 binary_search_2phase_cta), re-cut for 64 lanes: 64 evenly spaced PIVOTS of the longer list sit in
// the wave's LDS strip, a key first finds its segment among the pivots (6 LDS steps), then finishes in global memory
// inside that segment (log2(len / 64) dependent loads; none for lists of up to 64 ids, which the strip holds whole).
// A wave owns the edges of 16 consecutive rows per grab; heavy rows are not cut (the A/B is run at sizes where that
// tail does not dominate).  Same count as the hash-set kernel (tests); measured against it in bench.py / DESIGN 4.6.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GDN_BLOCK)
tc_bs_count_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, int32_t row_lo, int32_t row_hi,
                   unsigned *__restrict__ cursor, unsigned long long *__restrict__ total) {
  __shared__ vid_t s_piv[GDN_WAVES_PER_BLOCK][64];
  __shared__ unsigned long long s_red[GDN_WAVES_PER_BLOCK];
  const unsigned lane = gdn_lane(), w = threadIdx.x >> 6;
  vid_t *piv = s_piv[w];
  unsigned long long count = 0;
  for (;;) {
    unsigned u0 = 0;
    if (lane == 0) u0 = atomicAdd(cursor, 16u);
    u0 = (unsigned)row_lo + __shfl(u0, 0, 64);
    if (u0 >= (unsigned)row_hi) break;
    const unsigned u1 = u0 + 16u < (unsigned)row_hi ? u0 + 16u : (unsigned)row_hi;
    for (unsigned u = u0; u < u1; u++) {
      const eoff_t ub = rowptr[u], ue = rowptr[u + 1];
      const unsigned du = (unsigned)(ue - ub);
      if (du < 2u) continue;  // a single out-neighbour closes no triangle
      for (eoff_t e = ub; e < ue; e++) {
        const vid_t v = colidx[e];
        const eoff_t vb = rowptr[v], ve = rowptr[v + 1];
        const unsigned dv = (unsigned)(ve - vb);
        if (dv == 0u) continue;
        const bool u_short = du <= dv;
        const vid_t *__restrict__ S = colidx + (u_short ? ub : vb);
        const vid_t *__restrict__ Lg = colidx + (u_short ? vb : ub);
        const unsigned ns = u_short ? du : dv, nl = u_short ? dv : du;
        // pivot j = Lg[floor(j * nl / 64)]: pivot 0 is the first id, segments are [p_j, p_{j+1})
        piv[lane] = Lg[(unsigned)(((unsigned long long)lane * nl) >> 6)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned hits = 0;
        for (unsigned i = lane; i < ns; i += 64) {
          const vid_t key = S[i];
          if (key < piv[0]) continue;
          int lo = 0, hi = 63;  // largest j with piv[j] <= key
#pragma unroll
          for (int st = 0; st < 6; st++) {
            const int mid = (lo + hi + 1) >> 1;
            if (piv[mid] <= key) lo = mid;
            else hi = mid - 1;
          }
          if (piv[lo] == key) {
            hits++;
            continue;
          }
          unsigned a = (unsigned)(((unsigned long long)lo * nl) >> 6) + 1u;          // behind the pivot itself
          unsigned b = (unsigned)(((unsigned long long)(lo + 1) * nl) >> 6);         // exclusive: the next pivot's position
          if (b > nl) b = nl;
          while (a < b) {
            const unsigned mid = (a + b) >> 1;
            const vid_t x = Lg[mid];
            if (x == key) {
              hits++;
              break;
            }
            if (x < key) a = mid + 1u;
            else b = mid;
          }
        }
        count += hits;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the strip is refilled for the next edge
      }
    }
  }
  count = gdn_wave_sum(count);
  if (lane == 0) s_red[w] = count;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < GDN_WAVES_PER_BLOCK; i++) t += s_red[i];
    if (t) atomicAdd(total, t);
  }
}

int gdn_exclusive_scan_u32_to_u64(const uint32_t *d_in, eoff_t *d_out, size_t n, hipStream_t s);

// symmetric graph -> DAG (device arrays owned by the returned graph)
static int tc_orient(const gdn_graph *g, gdn_graph **out) {
  const int32_t m = g->m;
  DevBuf<int32_t> deg;
  DevBuf<unsigned> keep;
  DevBuf<eoff_t> pos;
  DevBuf<unsigned long long> bigitems;
  DevBuf<unsigned> cnt;
  const uint64_t bigcap64 = g->nnz / EXP_CHUNK + (uint64_t)m / 64 + 1024;
  const unsigned bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
  GDN_TRY(deg.alloc(m));
  GDN_TRY(keep.alloc(g->nnz));
  GDN_TRY(pos.alloc(g->nnz + 1));
  GDN_TRY(bigitems.alloc(bigcap));
  GDN_TRY(cnt.alloc(2));
  GDN_HIP(hipMemset(cnt.p, 0, 8));
  GDN_TRY(gdn_graph_degrees_dev(g, deg.p, nullptr));
  ExpBigList big;
  big.items = bigitems.p;
  big.capacity = bigcap;
  big.count = cnt.p;
  big.overflow = cnt.p + 1;
  TcKeepVis vis;
  vis.colidx = g->colidx;
  vis.deg = deg.p;
  vis.keep = keep.p;
  vis.v = 0;
  hipLaunchKernelGGL(tc_keep_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, big, vis);
  hipLaunchKernelGGL(tc_keep_big_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, vis);
  GDN_HIP(hipGetLastError());
  GDN_TRY(gdn_exclusive_scan_u32_to_u64(keep.p, pos.p, (size_t)g->nnz, 0));
  eoff_t nnz_dag = 0;
  GDN_HIP(hipMemcpy(&nnz_dag, pos.p + g->nnz, sizeof(eoff_t), hipMemcpyDeviceToHost));
  unsigned ovf[2];
  GDN_HIP(hipMemcpy(ovf, cnt.p, 8, hipMemcpyDeviceToHost));
  if (ovf[1]) {
    gdn_set_error("gdn_tc: device worklist overflow");
    return GDN_ERR_OVERFLOW;
  }
  gdn_graph *d = new gdn_graph();
  d->m = m;
  d->nnz = nnz_dag;
  d->owned = true;
  hipError_t e = gdn_plain_malloc((void **)&d->rowptr, ((size_t)m + 1) * sizeof(eoff_t));
  if (e == hipSuccess) e = gdn_plain_malloc((void **)&d->colidx, (nnz_dag ? nnz_dag : 1) * sizeof(vid_t));
  if (e != hipSuccess) {
    gdn_set_error("gdn_tc: %s", hipGetErrorString(e));
    gdn_graph_free(d);
    return GDN_ERR_OOM;
  }
  if (g->nnz) {
    size_t nb = (g->nnz + GDN_BLOCK - 1) / GDN_BLOCK;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(tc_compact_kernel, dim3((unsigned)nb), dim3(GDN_BLOCK), 0, 0, g->colidx, keep.p, pos.p,
                       g->nnz, d->colidx);
  }
  hipLaunchKernelGGL(tc_rowptr_kernel, dim3(gdn_nblocks((uint64_t)m + 1)), dim3(GDN_BLOCK), 0, 0, g->rowptr, pos.p, m,
                     d->rowptr);
  GDN_HIP(hipGetLastError());
  GDN_HIP(hipDeviceSynchronize());
  *out = d;
  return GDN_OK;
}

// ------------------------------------------------------------------------------------------
// The FORWARD form (default): vertices are relabelled by DEGREE RANK -- the order (degree, id) the orientation rule of
// src/common/graph.cc:80-81 compares by -- so that the DAG is simply "edges to higher ids" and every list is sorted by rank.
// For a DAG edge u -> v and the set N+(v), the elements of N+(u) that can be in the set are the ones BEHIND v in u's list (a
// member of N+(v) outranks v): the walk of N+(u) starts there.  Look-ups: SUM_u C(d+(u), 2) instead of SUM_u d+(u)^2 -- half
// of the v-centric count's, the "forward" algorithm.  The triangle count does not depend on the labelling.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GDN_BLOCK)
tc_indeg_kernel(const vid_t *__restrict__ colidx, uint64_t nnz, int32_t *__restrict__ deg) {
  for (size_t k = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; k < nnz; k += (size_t)gridDim.x * GDN_BLOCK) atomicAdd(&deg[colidx[k]], 1);
}
__global__ void __launch_bounds__(GDN_BLOCK)
tc_rank_keys_kernel(const int32_t *__restrict__ deg, int32_t m, unsigned long long *__restrict__ keys) {
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < (unsigned)m) keys[v] = ((unsigned long long)(unsigned)deg[v] << 32) | v;
}
__global__ void __launch_bounds__(GDN_BLOCK)
tc_newid_kernel(const unsigned long long *__restrict__ sorted, int32_t m, vid_t *__restrict__ newid) {
  const unsigned r = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (r < (unsigned)m) newid[(unsigned)(sorted[r] & 0xFFFFFFFFull)] = (vid_t)r;
}
struct TcRelabelVis {
  const vid_t *__restrict__ colidx;
  const vid_t *__restrict__ newid;
  unsigned long long *__restrict__ keys;  // one per CSR entry: (low rank << 32 | high rank), a self loop where nothing is kept
  int both;  // the input lists every edge in both directions (symmetric graph): keep the entry with rank(src) < rank(dst)
  unsigned *__restrict__ descends;  // oriented input: set when an edge does not ascend in rank (see tc_forward_build)
  int32_t v;
  __device__ __forceinline__ void begin_big(vid_t vv) { v = vv; }
  __device__ __forceinline__ void edge(int owner, eoff_t k, bool valid) {
    const int32_t src = __shfl(v, owner, 64);
    if (valid) {
      const unsigned a = (unsigned)newid[src], b = (unsigned)newid[colidx[k]];
      unsigned long long key;
      if (both) key = a < b ? (((unsigned long long)a << 32) | b) : (((unsigned long long)a << 32) | a);  // self loop = dropped
      else {
        key = a < b ? (((unsigned long long)a << 32) | b) : (((unsigned long long)b << 32) | a);
        if (a >= b) *descends = 1u;
      }
      keys[k] = key;
    }
  }
};
__global__ void __launch_bounds__(GDN_BLOCK)
tc_relabel_kernel(const eoff_t *__restrict__ rowptr, int32_t m, ExpBigList big, TcRelabelVis vis) {
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
tc_relabel_big_kernel(const eoff_t *__restrict__ rowptr, ExpBigList big, TcRelabelVis vis) {
  vis.v = 0;
  gdn_expand_big_items(rowptr, big, vis);
}
// nstart[j] for the in-edge slot j = (v <- u): 1 + the position of v in N+(u) (binary search; every list is ascending)
struct TcStartVis {
  const eoff_t *__restrict__ rowptr;   // the DAG
  const vid_t *__restrict__ colidx;
  const vid_t *__restrict__ in_col;    // its transpose
  unsigned *__restrict__ nstart;
  // nullable: the walk of slot j as ONE 8-byte word, (first element << 24) | elements -- the count kernel then reads it beside
  // the neighbour ids instead of gathering the neighbour's two row offsets (a dependent round trip per 64 neighbours less);
  // *nbound_bad is set when a walk does not fit the packing (2^24 elements, 2^40 offsets)
  unsigned long long *__restrict__ nbound;
  unsigned *__restrict__ nbound_bad;
  int32_t v;
  __device__ __forceinline__ void begin_big(vid_t vv) { v = vv; }
  __device__ __forceinline__ void edge(int owner, eoff_t j, bool valid) {
    const int32_t row = __shfl(v, owner, 64);
    if (valid) {
      const vid_t u = in_col[j];
      const eoff_t b = rowptr[u];
      const unsigned du = (unsigned)(rowptr[u + 1] - b);
      unsigned lo = 0, hi = du;
      while (lo < hi) {  // first position with an id > row
        const unsigned mid = (lo + hi) >> 1;
        if (colidx[b + mid] <= row) lo = mid + 1;
        else hi = mid;
      }
      nstart[j] = lo;
      if (nbound) {
        const unsigned long long first = b + lo, len = du - lo;
        if (len >= (1ull << 24) || first >= (1ull << 40)) *nbound_bad = 1u;
        nbound[j] = (first << 24) | (len & 0xFFFFFFull);
      }
    }
  }
};
__global__ void __launch_bounds__(GDN_BLOCK)
tc_start_kernel(const eoff_t *__restrict__ in_rowptr, int32_t m, ExpBigList big, TcStartVis vis) {
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK][64];
  const unsigned v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  eoff_t b = 0, e = 0;
  vis.v = (int32_t)v;
  if (v < (unsigned)m) {
    b = in_rowptr[v];
    e = in_rowptr[v + 1];
  }
  gdn_expand_wave(b, e, (vid_t)v, big, vis, s_scan[threadIdx.x >> 6]);
}
__global__ void __launch_bounds__(GDN_BLOCK)
tc_start_big_kernel(const eoff_t *__restrict__ in_rowptr, ExpBigList big, TcStartVis vis) {
  vis.v = 0;
  gdn_expand_big_items(in_rowptr, big, vis);
}

int gdn_radix_sort_u64(unsigned long long *a, unsigned long long *b, unsigned long long n, unsigned begin_bit, unsigned end_bit,
                       const unsigned long long **sorted);
int gdn_build_csr_from_keys(DevBuf<unsigned long long> &ka, DevBuf<unsigned long long> &kb, unsigned long long n, int32_t m,
                            gdn_graph **out);

// g: a symmetric graph (oriented == false) or its orientation by (degree, id) -- what the reference's
// Graph::orientation (src/common/graph.cc:67-113) hands to TCSolver -- (oriented == true) -> the rank-ordered DAG, its
// transpose and the walk starts of the forward count.
// The forward count is the number of triangles of the underlying simple graph; the reference's loop (src/tc/omp_base.cc:16-22)
// counts, on WHATEVER directed graph it is given, the pairs (u -> v, w in N+(u) and N+(v)).  The two agree when every edge
// of an oriented input ascends in the (degree, id) order and no entry repeats -- checked here, one flag and one count.
// Returns 1 (nothing built) when it does not hold: the caller then counts on the caller's orientation as it is.
static int tc_forward_build(const gdn_graph *g, bool oriented, gdn_graph **dag_out, gdn_graph **in_out, DevBuf<unsigned> &nstart,
                            DevBuf<unsigned long long> &nbound) {
  const int32_t m = g->m;
  DevBuf<int32_t> deg;
  DevBuf<vid_t> newid;
  DevBuf<unsigned long long> ra, rb, ka, kb, bigitems;
  DevBuf<unsigned> cnt;
  GDN_TRY(deg.alloc(m));
  GDN_TRY(newid.alloc(m));
  GDN_TRY(ra.alloc(m));
  GDN_TRY(rb.alloc(m));
  GDN_TRY(gdn_graph_degrees_dev(g, deg.p, nullptr));
  if (oriented && g->nnz)  // undirected degree = out + in
    hipLaunchKernelGGL(tc_indeg_kernel, dim3(4096), dim3(GDN_BLOCK), 0, 0, g->colidx, g->nnz, deg.p);
  hipLaunchKernelGGL(tc_rank_keys_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, deg.p, m, ra.p);
  GDN_HIP(hipGetLastError());
  const unsigned long long *sorted = nullptr;
  GDN_TRY(gdn_radix_sort_u64(ra.p, rb.p, (unsigned long long)m, 32u, 64u, &sorted));  // stable: ties keep the id order
  hipLaunchKernelGGL(tc_newid_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, sorted, m, newid.p);
  GDN_HIP(hipGetLastError());
  const uint64_t bigcap64 = g->nnz / EXP_CHUNK + (uint64_t)m / 64 + 1024;
  const unsigned bigcap = (unsigned)(bigcap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : bigcap64);
  GDN_TRY(ka.alloc_scratch(g->nnz));
  GDN_TRY(kb.alloc_scratch(g->nnz));
  GDN_TRY(bigitems.alloc(bigcap));
  GDN_TRY(cnt.alloc(4));
  GDN_HIP(hipMemset(cnt.p, 0, 16));
  ExpBigList big;
  big.items = bigitems.p;
  big.capacity = bigcap;
  big.count = cnt.p;
  big.overflow = cnt.p + 1;
  TcRelabelVis rv;
  rv.colidx = g->colidx;
  rv.newid = newid.p;
  rv.keys = ka.p;
  rv.both = oriented ? 0 : 1;
  rv.descends = cnt.p + 2;
  rv.v = 0;
  hipLaunchKernelGGL(tc_relabel_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, g->rowptr, m, big, rv);
  hipLaunchKernelGGL(tc_relabel_big_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, g->rowptr, big, rv);
  GDN_HIP(hipGetLastError());
  unsigned ovf[4];
  GDN_HIP(hipMemcpy(ovf, cnt.p, 16, hipMemcpyDeviceToHost));
  if (ovf[1]) {
    gdn_set_error("gdn_tc: device worklist overflow");
    return GDN_ERR_OVERFLOW;
  }
  if (oriented && ovf[2]) return 1;  // not the (degree, id) orientation: count on the input as it is
  deg.release();
  ra.release();
  rb.release();
  gdn_graph *dag = nullptr, *din = nullptr;
  GDN_TRY(gdn_build_csr_from_keys(ka, kb, g->nnz, m, &dag));
  if (oriented && dag->nnz != g->nnz) {  // repeated entries (or self loops) in an oriented input: the reference counts them
    gdn_graph_free(dag);
    return 1;
  }
  int rc = dag->nnz ? gdn_graph_transpose(dag, &din) : GDN_OK;
  if (rc == GDN_OK && din) {
    rc = nstart.alloc(din->nnz);
    // GDN_TC_NBOUND=0: without the packed walk bounds (A/B); default from 2^26 DAG edges on (8 bytes per edge; RMAT-23 / 24 -1 .. -3 %, RMAT-22 the same)
    const char *enb = gdn_test_option("GDN_TC_NBOUND");
    const bool want_nbound = enb ? enb[0] != '0' : din->nnz >= (1ull << 26);
    DevBuf<unsigned> nb_bad;
    if (rc == GDN_OK && want_nbound) {
      rc = nbound.alloc(din->nnz);
      if (rc == GDN_OK) rc = nb_bad.alloc(1);
      if (rc == GDN_OK && hipMemset(nb_bad.p, 0, 4) != hipSuccess) rc = GDN_ERR_HIP;
    }
    if (rc == GDN_OK && hipMemset(cnt.p, 0, 8) != hipSuccess) {
      gdn_set_error("gdn_tc: hipMemset failed");
      rc = GDN_ERR_HIP;
    }
    if (rc == GDN_OK) {
      TcStartVis sv;
      sv.rowptr = dag->rowptr;
      sv.colidx = dag->colidx;
      sv.in_col = din->colidx;
      sv.nstart = nstart.p;
      sv.nbound = nbound.p;
      sv.nbound_bad = nb_bad.p;
      sv.v = 0;
      hipLaunchKernelGGL(tc_start_kernel, dim3(gdn_nblocks((uint64_t)m)), dim3(GDN_BLOCK), 0, 0, din->rowptr, m, big, sv);
      hipLaunchKernelGGL(tc_start_big_kernel, dim3(1024), dim3(GDN_BLOCK), 0, 0, din->rowptr, big, sv);
      if (hipMemcpy(ovf, cnt.p, 8, hipMemcpyDeviceToHost) != hipSuccess || ovf[1]) {
        gdn_set_error("gdn_tc: start kernel failed or work list overflow");
        rc = GDN_ERR_OVERFLOW;
      }
      if (rc == GDN_OK && nbound.p) {
        unsigned bad = 0;
        if (hipMemcpy(&bad, nb_bad.p, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = GDN_ERR_HIP;
        else if (bad) nbound.release();  // a walk that does not fit the packing: the kernel gathers the row offsets as before
      }
    }
  }
  if (rc != GDN_OK) {
    if (din) gdn_graph_free(din);
    gdn_graph_free(dag);
    return rc;
  }
  *dag_out = dag;
  *in_out = din;
  return GDN_OK;
}

// ------------------------------------------------------------------------------------------
// The CORE of the forward count (round 4).  On a skewed graph most look-ups concern the few thousand vertices of highest
// rank: symmetrized RMAT-22, rank-ordered DAG -- 59 % of the forward count's look-ups have their middle vertex v among the
// top 8192 ranks (0.2 % of the vertices), 72 % among the top 16384 (numbers: DESIGN 4.7).  A member of N+(v) outranks v, so
// for such a v every candidate lies in the same K ranks: the adjacency AMONG the top K ranks is kept as a K x K BIT MATRIX
// (row v = N+(v); 8 MB at K = 8192: L2 / MALL resident), and a look-up becomes one bit of a row AND.
//   for every u:  C(u) = N+(u) restricted to the core (a suffix of the ascending list), B_u = its bitmap (one wave: K / 64
//                 bits per lane, in registers);  triangles(u, v in core) = SUM over v in C(u) of popcount(row_v AND B_u)
// -- row_v holds only ranks above v, so every pair (v < w) of C(u) is met once.  A step of the hash-set kernel settles 64
// look-ups with a 256-byte list load, 64 bucket reads and ~25 instructions; a row AND settles K candidates with K / 8 bytes
// of a coalesced, cached row and ~6 instructions per 4096 of them.  Rows are only read between the word of v itself and the
// word of C(u)'s last member.  The u with fewer than TC_CORE_SMALL core neighbours test their C(|C|, 2) pairs bit by bit
// instead (a row per v would be mostly zeros of B_u): 0.6 % of the look-ups.
// The rows v below the core stay with tc_count_kernel (its row range ends where the core begins): the two parts add up to
// the forward count, whatever K is.
// ------------------------------------------------------------------------------------------
#define TC_CORE_SMALL 32  // |C(u)| below this: pair tests (must be <= 64: one id per lane)

// one wave per core row: its list as bits
__global__ void __launch_bounds__(GDN_BLOCK)
tc_core_adj_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, unsigned base, unsigned k_core, unsigned kw,
                   unsigned long long *__restrict__ adj) {
  const unsigned lane = gdn_lane();
  const size_t nwaves = ((size_t)gridDim.x * GDN_BLOCK) >> 6;
  for (size_t i = ((size_t)blockIdx.x * GDN_BLOCK + threadIdx.x) >> 6; i < (size_t)k_core; i += nwaves) {
    const eoff_t b = rowptr[base + i], e = rowptr[base + i + 1];
    for (eoff_t k = b + lane; k < e; k += 64) {
      const unsigned j = (unsigned)colidx[k] - base;  // > i: lists ascend in rank
      atomicOr(&adj[i * kw + (j >> 6)], 1ull << (j & 63u));
    }
  }
}
// the rows with at least two core neighbours: item = length class << 62 | |C(u)| << 32 | u (C(u) is the END of the row);
// cls[c] = items with |C(u)| >= tc_core_class_min(c) -- the list is partitioned by class afterwards and handed out longest
// class first, see tc_core_count_kernel (a full sort by |C(u)| was measured: it scatters the rows of a grab over the whole
// column array, 7.0 -> 7.8 ms at K = 4096)
#define TC_CORE_CLASSES 4
#define TC_CORE_CUR 64  // word of the control block where the classes' work counters start (TC_NCUR per class)
#define TC_CORE_CTL_WORDS (TC_CORE_CUR + TC_CORE_CLASSES * TC_NCUR * TC_CUR_STRIDE)
__device__ __host__ constexpr unsigned tc_core_class_min(int c) { return c == 0 ? 256u : c == 1 ? 64u : c == 2 ? 16u : 2u; }
#ifndef TC_CORE_TAKE_SHIFT
#define TC_CORE_TAKE_SHIFT 0
#endif
__device__ __host__ constexpr unsigned tc_core_class_take(int c) {
  return c == 0 ? 1u : (c == 1 ? 4u : c == 2 ? 16u : 64u) >> TC_CORE_TAKE_SHIFT;
}
__global__ void __launch_bounds__(GDN_BLOCK)
tc_core_items_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, int32_t m, unsigned base,
                     unsigned long long *__restrict__ items, unsigned *__restrict__ n_items, unsigned *__restrict__ cls) {
  const unsigned u = blockIdx.x * GDN_BLOCK + threadIdx.x;
  unsigned n = 0;
  if (u < (unsigned)m) {
    const eoff_t b = rowptr[u];
    const unsigned d = (unsigned)(rowptr[u + 1] - b);
    unsigned lo = 0, hi = d;
    while (lo < hi) {  // first position with a core id
      const unsigned mid = (lo + hi) >> 1;
      if ((unsigned)colidx[b + mid] < base) lo = mid + 1;
      else hi = mid;
    }
    n = d - lo;
  }
  const bool keep = n >= 2u;
  const unsigned long long mask = __ballot(keep);
  if (mask == 0ull) return;
  unsigned at = 0, kc[TC_CORE_CLASSES - 1];
#pragma unroll
  for (int c = 0; c + 1 < TC_CORE_CLASSES; c++) kc[c] = (unsigned)__popcll(__ballot(n >= tc_core_class_min(c)));
  if (gdn_lane() == 0) {
    at = atomicAdd(n_items, (unsigned)__popcll(mask));
#pragma unroll
    for (int c = 0; c + 1 < TC_CORE_CLASSES; c++)
      if (kc[c]) atomicAdd(&cls[c], kc[c]);
  }
  at = __shfl(at, 0, 64) + (unsigned)__popcll(mask & gdn_lanemask_lt());
  if (keep) {  // bits 62..63: 3 - class (the list is then partitioned by these two bits, stable: rows stay in order inside a class)
    const unsigned code = n >= tc_core_class_min(0) ? 3u : n >= tc_core_class_min(1) ? 2u : n >= tc_core_class_min(2) ? 1u : 0u;
    items[at] = ((unsigned long long)code << 62) | ((unsigned long long)n << 32) | u;
  }
}

// List elements the forward count WALKS with a core of the ranks >= base: the look-ups around a middle vertex v < base.  Row u
// with d out-neighbours of which the first n lie below base contributes SUM_{i < n} (d - 1 - i) = n (d - 1) - n (n - 1) / 2
// (edge i of the ascending list starts its walk behind itself); the look-ups around v >= base are row ANDs of the bit matrix.
__global__ void __launch_bounds__(GDN_BLOCK)
tc_walked_elements_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, int32_t m, unsigned base,
                          unsigned long long *__restrict__ out) {
  __shared__ unsigned long long s_red[GDN_WAVES_PER_BLOCK];
  unsigned long long acc = 0;
  for (uint64_t u = (uint64_t)blockIdx.x * GDN_BLOCK + threadIdx.x; u < (uint64_t)m; u += (uint64_t)gridDim.x * GDN_BLOCK) {
    const eoff_t b = rowptr[u];
    const unsigned long long d = rowptr[u + 1] - b;
    unsigned long long lo = 0, hi = d;
    while (lo < hi) {  // first position with an id >= base
      const unsigned long long mid = (lo + hi) >> 1;
      if ((unsigned)colidx[b + mid] < base) lo = mid + 1;
      else hi = mid;
    }
    if (lo) acc += lo * (d - 1) - lo * (lo - 1) / 2;
  }
  acc = gdn_block_sum(acc, s_red);
  if (threadIdx.x == 0 && acc) atomicAdd(out, acc);
}

typedef unsigned tc_u32x2 __attribute__((ext_vector_type(2)));
#ifndef TC_CORE_WPE
#define TC_CORE_WPE 8  // <= 64 vector registers: two waves of this kernel fit a SIMD beside four of tc_count_kernel's (see TC_WAVES_PER_EU)
#endif
template <int R>  // K = 4096 R: R 64-bit words of a row (and of B_u) per lane
__global__ void __launch_bounds__(GDN_BLOCK, TC_CORE_WPE)
tc_core_count_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, const unsigned long long *__restrict__ items,
                     unsigned *__restrict__ ctl, const unsigned long long *__restrict__ adj, unsigned base,
                     unsigned long long *__restrict__ total, unsigned small) {
  constexpr unsigned KW = 64u * R;
  // rows per group: 4 / 2 / 2 / 1 -- the kernel runs in the registers tc_count_kernel leaves free (128 per lane and SIMD beside its
  // 4 x 96: two waves of <= 64; rounds 4-5, 160 free beside 4 x 88: R = 4 with four rows per group, 110 registers, ran one wave per
  // SIMD: 27 ms beside the hash-set kernel where alone it takes 12).  Two groups are in flight either way.  Round 6, once the register budget beside the hash-set kernel was understood (two
  // waves of <= 64): K = 12288 with TWO rows per group (64 registers, 5 spilled) RMAT-23 10.80 -> 10.17 ms -- with 70 registers
  // and no bound only one wave fits: 11.5; K = 8192 with three (64, 2 spilled): the same as two; K = 16384 with two (20 spilled):
  // RMAT-24 28.8 -> 29.3; K = 4096 with six: the same (sessions r06_69, r06_70)
#ifndef TC_CORE_G3
#define TC_CORE_G3 2
#endif
#ifndef TC_CORE_G2
#define TC_CORE_G2 2
#endif
#ifndef TC_CORE_G4
#define TC_CORE_G4 1
#endif
#ifndef TC_CORE_G1
#define TC_CORE_G1 4
#endif
  constexpr int G = R >= 4 ? TC_CORE_G4 : R == 3 ? TC_CORE_G3 : R == 2 ? TC_CORE_G2 : TC_CORE_G1;
  __shared__ unsigned long long s_bm[GDN_WAVES_PER_BLOCK][KW];
  __shared__ unsigned long long s_red[GDN_WAVES_PER_BLOCK];
  const unsigned lane = gdn_lane(), w = threadIdx.x >> 6;
  const unsigned *__restrict__ adj32 = reinterpret_cast<const unsigned *>(adj);
  unsigned long long count = 0;
  // The list is partitioned by length class, shortest class first, and handed out from its END, class by class (ctl: [0]
  // items, [1 + c] the cursor of class c, [5 + c] the items of at least class c's length): 1 item per grab among the
  // longest, 64 among the shortest -- one atomic on a shared cursor per short item costs more than the item (a hot address
  // serves ~50 M atomics / s), 64 long items in one grab are the kernel's tail.  A grab's items and row ends are loaded by
  // its lanes side by side.
  const unsigned n_items = ctl[0];
  const unsigned gw = blockIdx.x * GDN_WAVES_PER_BLOCK + w;
  int cls = 0;  // (wave-uniform)
  TcGrab grab(ctl + TC_CORE_CUR, gw);  // a class's grabs go over TC_NCUR counters like tc_count_kernel's (round 6)
  for (;;) {
    unsigned it0 = 0, lim = 0, take = 0;
    for (; cls < TC_CORE_CLASSES; cls++) {
      const unsigned first = cls == 0 ? 0u : ctl[5 + cls - 1];
      lim = cls + 1 < TC_CORE_CLASSES ? ctl[5 + cls] : n_items;
      take = tc_core_class_take(cls);
      if (first < lim) {
        const unsigned unit = grab.next((lim - first + take - 1u) / take, lane);
        if (unit != ~0u) {
          it0 = first + unit * take;
          break;
        }
      }
      grab = TcGrab(ctl + TC_CORE_CUR + (unsigned)(cls + 1) * (TC_NCUR * TC_CUR_STRIDE), gw);
    }
    if (cls >= TC_CORE_CLASSES) break;
    take = lim - it0 < take ? lim - it0 : take;
    eoff_t ej = 0;
    unsigned nj = 0;
    if (lane < take) {
      const unsigned long long item = items[n_items - 1u - (it0 + lane)];
      nj = (unsigned)(item >> 32) & 0x3FFFFFFFu;
      ej = rowptr[(unsigned)(item & 0xFFFFFFFFull) + 1u];
    }
    for (unsigned jt = 0; jt < take; jt++) {
      const unsigned n = (unsigned)__builtin_amdgcn_readlane((int)nj, (int)jt);
      const eoff_t b = (((eoff_t)(unsigned)__builtin_amdgcn_readlane((int)(ej >> 32), (int)jt) << 32) |
                        (unsigned)__builtin_amdgcn_readlane((int)ej, (int)jt)) - n;  // C(u) is the end of the row
      unsigned cnt = 0;
      if (n < small) {
        // pairs (i < j): lane j tests bit c_j of row c_i, four rows in flight
        const unsigned cj = lane < n ? (unsigned)colidx[b + lane] - base : 0u;
        for (unsigned i0 = 0; i0 + 1u < n; i0 += 4u) {
          unsigned wd[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const unsigned i = i0 + (unsigned)q;
            const unsigned ci = (unsigned)__builtin_amdgcn_readlane((int)cj, (int)(i < 63u ? i : 63u));
            wd[q] = (lane > i && lane < n) ? adj32[(size_t)ci * (2u * KW) + (cj >> 5)] : 0u;
          }
#pragma unroll
          for (int q = 0; q < 4; q++) cnt += (wd[q] >> (cj & 31u)) & 1u;
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; r++) s_bm[w][lane + 64u * r] = 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned c_last = 0;
        for (unsigned k = lane; k < n; k += 64u) {
          const unsigned c = (unsigned)colidx[b + k] - base;
          atomicOr(&s_bm[w][c >> 6], 1ull << (c & 63u));
          c_last = c;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned long long B[R];
#pragma unroll
        for (int r = 0; r < R; r++) B[r] = s_bm[w][lane + 64u * r];
        // words behind the one of C(u)'s last member are zero in B_u: not read
        const unsigned w_hi = (unsigned)__builtin_amdgcn_readlane((int)c_last, (int)((n - 1u) & 63u)) >> 6;
        // A row is read through a BUFFER descriptor of its useful words [w_lo, w_hi] (scalar registers): lanes whose word
        // lies outside get zero from the bounds check, without a branch and without a memory access -- with predicated loads
        // every load sat in its own basic block and a group of rows cost a full round trip before the next was issued.
        // G rows per group, the next group's loads issued before this group's popcounts.
        for (unsigned k0 = 0; k0 + 1u < n; k0 += 64u) {
          const unsigned ck = k0 + lane < n ? (unsigned)colidx[b + k0 + lane] - base : 0u;
          const unsigned kn = n - 1u - k0 < 64u ? n - 1u - k0 : 64u;  // (the last member closes nothing)
          auto issue = [&](unsigned kk, tc_u32x2 (&x)[G][R]) {
#pragma unroll
            for (int q = 0; q < G; q++) {
              const unsigned k = kk + (unsigned)q;
              const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)ck, (int)(k < 63u ? k : 63u));
              const unsigned w_lo = c >> 6;  // row c holds ranks above c only
              const unsigned bytes = (k < kn && w_hi >= w_lo) ? (w_hi - w_lo + 1u) * 8u : 0u;
              const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                  const_cast<unsigned long long *>(adj + (size_t)c * KW + w_lo), (short)0, (int)bytes, 0x00020000);
              const unsigned off = (lane - w_lo) * 8u;  // (wraps for the words below w_lo: out of range)
#pragma unroll
              for (int r = 0; r < R; r++) x[q][r] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(off + 512u * (unsigned)r), 0, 0);
            }
          };
          auto consume = [&](const tc_u32x2 (&x)[G][R]) {
#pragma unroll
            for (int q = 0; q < G; q++)
#pragma unroll
              for (int r = 0; r < R; r++)
                cnt += (unsigned)__popc(x[q][r].x & (unsigned)B[r]) + (unsigned)__popc(x[q][r].y & (unsigned)(B[r] >> 32));
          };
          tc_u32x2 x0[G][R], x1[G][R];
          issue(0u, x0);
          for (unsigned kk = 0; kk < kn; kk += 2u * G) {
            issue(kk + (unsigned)G, x1);
            consume(x0);
            issue(kk + 2u * G, x0);
            consume(x1);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // B_u is rebuilt for the next item
      }
      count += cnt;
    }
  }
  count = gdn_wave_sum(count);
  if (lane == 0) s_red[w] = count;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < GDN_WAVES_PER_BLOCK; i++) t += s_red[i];
    if (t) atomicAdd(total, t);
  }
}

// SURVEY 8d's merge-equivalent traffic of a count: SUM over DAG edges (u,v) of d+(u) + d+(v) (a merge intersect reads both lists)
__global__ void __launch_bounds__(GDN_BLOCK)
tc_model_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, int32_t m, unsigned long long *__restrict__ out) {
  // out[0] += SUM d+(u) + d+(v); out[1] += SUM d+(u) alone (the probes of the v-centric count; the rest are the u-centric's)
  unsigned long long acc = 0, acc_u = 0;
  const unsigned lane = gdn_lane();
  const size_t wave = ((size_t)blockIdx.x * GDN_BLOCK + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * GDN_BLOCK) >> 6;
  for (size_t u = wave; u < (size_t)m; u += nwaves) {
    const eoff_t a = rowptr[u], b = rowptr[u + 1];
    const unsigned long long du = b - a;
    for (eoff_t e = a + lane; e < b; e += 64) {
      const vid_t v = colidx[e];
      acc += du + (rowptr[v + 1] - rowptr[v]);
      acc_u += du;
    }
  }
  acc = gdn_wave_sum(acc);
  acc_u = gdn_wave_sum(acc_u);
  if (lane == 0 && acc) {
    atomicAdd(out, acc);
    atomicAdd(out + 1, acc_u);
  }
}

extern "C" {

// bytes = 4 * SUM_{(u,v) in DAG} (d+(u) + d+(v)) + 4 nnz [src list] + 4 nnz [colidx] + 8 (m + 1)   (TC row)
int gdn_tc_model_bytes(const gdn_graph *dag, uint64_t *bytes) {
  GDN_REQUIRE(dag != nullptr && bytes != nullptr, "dag / bytes");
  DevBuf<unsigned long long> acc;
  GDN_TRY(acc.alloc(2));
  GDN_HIP(hipMemset(acc.p, 0, 16));
  hipLaunchKernelGGL(tc_model_kernel, dim3(4096), dim3(GDN_BLOCK), 0, 0, dag->rowptr, dag->colidx, dag->m, acc.p);
  unsigned long long h[2] = {0, 0};
  GDN_HIP(hipMemcpy(h, acc.p, 16, hipMemcpyDeviceToHost));
  *bytes = 4ull * h[0] + 8ull * dag->nnz + 8ull * ((uint64_t)dag->m + 1);
  return GDN_OK;
}

// the probes of the two formulations of tc_count_kernel: probes[0] u-centric (SUM d+(v)), probes[1] v-centric (SUM d+(u))
static int tc_probe_counts(const gdn_graph *dag, unsigned long long probes[2]) {
  DevBuf<unsigned long long> acc;
  GDN_TRY(acc.alloc(2));
  GDN_HIP(hipMemset(acc.p, 0, 16));
  hipLaunchKernelGGL(tc_model_kernel, dim3(4096), dim3(GDN_BLOCK), 0, 0, dag->rowptr, dag->colidx, dag->m, acc.p);
  unsigned long long h[2] = {0, 0};
  GDN_HIP(hipMemcpy(h, acc.p, 16, hipMemcpyDeviceToHost));
  probes[0] = h[0] - h[1];
  probes[1] = h[1];
  return GDN_OK;
}

int gdn_tc_probe_counts(const gdn_graph *dag, uint64_t *probes) {
  GDN_REQUIRE(dag != nullptr && probes != nullptr, "dag / probes");
  unsigned long long p[2] = {0, 0};
  GDN_TRY(tc_probe_counts(dag, p));
  probes[0] = p[0];
  probes[1] = p[1];
  return GDN_OK;
}

// triangles closed over the source rows [row_lo, row_hi) of an oriented graph (the light-row cursor starts at row_lo and
// the kernel's vertex bound is row_hi: the count kernel itself does not know about ranges)
// dag_in != nullptr: the v-centric count over the rows [row_lo, row_hi) of the TRANSPOSED DAG's row space (same vertices)
struct gdn_tc_plan;
static int tc_core_prepare(gdn_tc_plan &p);           // cursors and total zeroed on the null stream, the core stream waits for that
static int tc_core_launch(gdn_tc_plan &p, bool tail);  // tail: the grid that takes what is left once tc_count_kernel is done
// the heavy rows' work items of a plan's count: made by its first count, kept (they depend on the DAG and the limit only)
static int tc_plan_heavy_items(gdn_tc_plan &p, const gdn_graph *dag, const gdn_graph *nb_graph, int32_t row_lo, int32_t row_hi,
                               unsigned light, unsigned cap, unsigned long long **items, unsigned **ctl);
static int tc_count_rows(const gdn_graph *dag, int32_t row_lo, int32_t row_hi, uint64_t *total, gdn_stats &st,
                         const gdn_graph *dag_in = nullptr, bool binary_search = false, const unsigned *nstart = nullptr,
                         gdn_tc_plan *core = nullptr /* its core kernel is queued right behind tc_count_kernel's launch */,
                         const unsigned long long *nbound = nullptr) {
  const gdn_graph *nb_graph = dag_in ? dag_in : dag;  // where a row's neighbours come from
  DevBuf<unsigned long long> d_total, d_items;  // triangle count; (slice << 32 | row) items of the heavy rows
  DevBuf<unsigned> d_ctl;                       // [2] #items, [3] overflow, from [64]: the work counters of tc_count_kernel (TcGrab)
  const uint64_t cap64 = dag->nnz / TC_LIGHT_MIN + 1024;  // a heavy row of du > light >= TC_LIGHT_MIN ids yields ceil(du / TC_SLICE) <= du / TC_LIGHT_MIN items
  const unsigned cap = (unsigned)(cap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : cap64);
  HostTimer tprep, tsolve;
  tprep.start();
  GDN_TRY(d_total.alloc(1));
  if (!core) GDN_TRY(d_items.alloc(cap));
  GDN_TRY(d_ctl.alloc(64 + 2 * TC_NCUR * TC_CUR_STRIDE));  // (tc_bs_count_kernel: [0] its cursor)
  GDN_HIP(hipMemset(d_total.p, 0, 8));
  GDN_HIP(hipMemset(d_ctl.p, 0, (64 + 2 * TC_NCUR * TC_CUR_STRIDE) * sizeof(unsigned)));
  st.prep_ms += tprep.stop_ms();
  *total = 0;
  if (row_hi <= row_lo) return GDN_OK;
  tsolve.start();  // src/tc/gpu_base.cu:52-58
  const uint64_t rows = (uint64_t)(row_hi - row_lo);
  if (binary_search) {  // GDN_TC_FORM=bs: wave-per-edge binary-search intersect (tc_bs_count_kernel)
    GDN_HIP(hipMemset(d_ctl.p, 0, 16));
    hipLaunchKernelGGL(tc_bs_count_kernel, dim3(256 * 8), dim3(GDN_BLOCK), 0, 0, dag->rowptr, dag->colidx, row_lo, row_hi, d_ctl.p,
                       d_total.p);
    unsigned long long hb = 0;
    if (hipMemcpy(&hb, d_total.p, 8, hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_tc: binary-search count kernel failed: %s", hipGetErrorString(hipGetLastError()));
      return GDN_ERR_HIP;
    }
    st.solve_ms = tsolve.stop_ms();
    *total = hb;
    st.iterations = 1;
    return GDN_OK;
  }
  unsigned light = dag->m >= (1 << 21) ? 512u : 256u;  // rows up to this many neighbours: one wave, whole (see TC_LIGHT_MIN)
  // beside the core kernel (its rows are gone, the tail is what is left): RMAT-23 512 / 256 / 128 -> 21.7 / 21.3 / 20.7 ms
  if (core) light = 128u;
  if (const char *e = gdn_xoption("GDN_TC_LIGHT")) light = std::max((unsigned)atoi(e), (unsigned)TC_LIGHT_MIN);  // tuning knob
  if (core) GDN_TRY(tc_core_prepare(*core));
  unsigned long long *items_p = d_items.p;
  unsigned *ictl = d_ctl.p;  // [2] #items, [3] overflow
  if (core) {
    GDN_TRY(tc_plan_heavy_items(*core, dag, nb_graph, row_lo, row_hi, light, cap, &items_p, &ictl));
  } else {
    hipLaunchKernelGGL(tc_heavy_items_kernel, dim3(gdn_nblocks(rows)), dim3(GDN_BLOCK), 0, 0, dag->rowptr, nb_graph->rowptr, row_lo,
                       row_hi, items_p, cap, ictl + 2, ictl + 3, light);
  }
  unsigned nb = gdn_nblocks(rows, GDN_WAVES_PER_BLOCK * 16);
  if (nb > 256 * 8) nb = 256 * 8;  // persistent: up to 8 workgroups per CU pulling work items
  const size_t tc_lds = sizeof(vid_t) * GDN_WAVES_PER_BLOCK * TC_HASH + 8 * GDN_WAVES_PER_BLOCK + (size_t)GDN_WAVES_PER_BLOCK * 64 * TC_UNR;
  hipLaunchKernelGGL(tc_count_kernel, dim3(nb), dim3(GDN_BLOCK), tc_lds, 0, dag->rowptr, dag->colidx, nb_graph->rowptr,
                     nb_graph->colidx, row_hi, items_p, ictl + 2, d_ctl.p + 64, d_total.p, light, nstart, nbound, row_lo);
  if (core) {
    GDN_TRY(tc_core_launch(*core, false));  // beside tc_count_kernel, on the core's stream
    GDN_TRY(tc_core_launch(*core, true));   // behind it, on this stream
  }
  unsigned long long h = 0;
  unsigned ctl[4] = {0, 0, 0, 0};
  if (hipMemcpy(&h, d_total.p, 8, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(ctl, ictl, 16, hipMemcpyDeviceToHost) != hipSuccess) {
    gdn_set_error("gdn_tc: count kernel failed: %s", hipGetErrorString(hipGetLastError()));
    return GDN_ERR_HIP;
  }
  if (ctl[3]) {
    gdn_set_error("gdn_tc: heavy-row work list overflow");
    return GDN_ERR_OVERFLOW;
  }
  st.solve_ms = tsolve.stop_ms();
  *total = h;
  st.iterations = 1;
  return GDN_OK;
}

struct gdn_tc_plan {
  gdn_graph *dag = nullptr;     // the DAG the count runs on (owned): rank-ordered (forward) or the reference's orientation
  gdn_graph *dag_in = nullptr;  // its transpose (forward / v-centric), owned
  DevBuf<unsigned> nstart;      // forward: walk starts, parallel to dag_in->colidx
  DevBuf<unsigned long long> nbound;  // forward: (first element << 24 | elements) of every walk, parallel to dag_in->colidx (may be absent)
  int form = 0;                 // 0 u-centric, 1 v-centric, 2 binary search, 3 forward
  // forward: the core (tc_core_count_kernel): the top core_k ranks as a bit matrix, the rows with >= 2 core neighbours
  unsigned core_k = 0;                  // 0: no core
  DevBuf<unsigned long long> core_adj;  // core_k x core_k bits
  DevBuf<unsigned long long> core_items_a, core_items_b;  // the items, sorted by |C(u)| (the radix sort leaves them in one of the two)
  const unsigned long long *core_items = nullptr;
  DevBuf<unsigned> core_ctl;            // [0] items, [5..7] items of at least a class's length, from [TC_CORE_CUR]: the classes' work counters
  DevBuf<unsigned long long> core_total;
  hipStream_t core_stream = nullptr;    // the core kernel runs BESIDE tc_count_kernel (which fills half of a CU's wave slots)
  hipEvent_t core_ready = nullptr;      // cursors zeroed (null stream) -> the core's stream may start
  std::mutex count_mu;                  // the core's cursors and total belong to the plan: one count at a time
  // the heavy rows' work items of tc_count_kernel (tc_plan_heavy_items): [2] of the control words = their number, [3] = overflow
  DevBuf<unsigned long long> items;
  DevBuf<unsigned> items_ctl;
  unsigned items_light = 0;
  int32_t items_lo = -1, items_hi = -1;
  double prep_ms = 0;
  ~gdn_tc_plan() {
    if (core_stream) (void)hipStreamDestroy(core_stream);
    if (core_ready) (void)hipEventDestroy(core_ready);
    if (dag_in) gdn_graph_free(dag_in);
    if (dag) gdn_graph_free(dag);
  }
};

static int tc_plan_heavy_items(gdn_tc_plan &p, const gdn_graph *dag, const gdn_graph *nb_graph, int32_t row_lo, int32_t row_hi,
                               unsigned light, unsigned cap, unsigned long long **items, unsigned **ctl) {
  if (!p.items.p || p.items_light != light || p.items_lo != row_lo || p.items_hi != row_hi) {
    GDN_TRY(p.items.alloc(cap));
    GDN_TRY(p.items_ctl.alloc(4));
    GDN_HIP(hipMemsetAsync(p.items_ctl.p, 0, 16, 0));
    hipLaunchKernelGGL(tc_heavy_items_kernel, dim3(gdn_nblocks((uint64_t)(row_hi - row_lo))), dim3(GDN_BLOCK), 0, 0, dag->rowptr,
                       nb_graph->rowptr, row_lo, row_hi, p.items.p, cap, p.items_ctl.p + 2, p.items_ctl.p + 3, light);
    GDN_HIP(hipGetLastError());
    p.items_light = light;
    p.items_lo = row_lo;
    p.items_hi = row_hi;
  }
  *items = p.items.p;
  *ctl = p.items_ctl.p;
  return GDN_OK;
}

// GDN_TC_CORE: ranks of the core (4096, 8192, 12288 or 16384; 0 = none).  Default from 2^21 vertices on, 8192 / 12288 / 16384
// ranks from 2^21 / 2^22 / 2^24 vertices (round 4: 16384 throughout, on the numbers that follow): symmetrized
// R-MAT, count beside the hash-set kernel, K = 0 / 8192 / 12288 / 16384 -- scale 21: 7.2 / 7.2 / 6.8 / 6.5 ms, 22: 15.4 / 15.6 /
// 13.9 / 12.3, 23: 32.0 / 25.9 / 20.9 / 21.5, 24: 84.5 / 70.3 / 58.1 / 53.4 (profiles/r04_tc_core.txt).
static int tc_core_build(gdn_tc_plan &p) {
  const gdn_graph *dag = p.dag;
  // (round 5, final binary, profiles/r05_tc_core_k.txt: K = 8192 / 12288 / 16384 -- RMAT-21 5.29 / 5.77 / 5.92 ms, Orkut-like stand-in
  // (3.97 M vertices) 11.06 / 11.10 / 11.8, RMAT-22 10.5 / 9.63 / 10.1, RMAT-23 25.2 / 20.45 / 20.9, RMAT-24 - / 52.2 / 51.8)
  // (round 6: 8192 ranks already from 2^19 vertices -- RMAT-19 / 20 without / 4096 / 8192: 1.47 / 0.96 / 0.92 and 2.87 / 1.52 / 1.29 ms)
  // (session r06_56, the two kernels sharing a CU as they do now: RMAT-22 8192 / 12288 / 16384 4.61 / 5.12 / 5.61 ms, RMAT-23 15.1 / 11.0 / 11.5,
  // RMAT-24 - / 32.2 / 29.1, RMAT-21 4096 / 8192 / 12288 2.50 / 2.10 / 2.53: 12288 ranks from 2^23 vertices on)
  unsigned k = dag->m >= (1 << 24) ? 16384u : dag->m >= (1 << 23) ? 12288u : dag->m >= (1 << 19) ? 8192u : 0u;
  if (const char *e = gdn_option("GDN_TC_CORE")) k = (unsigned)atoi(e);
  k = k >= 16384u ? 16384u : (k / 4096u) * 4096u;  // whole lanes x 64 bits: 4096, 8192, 12288 or 16384
  if (k == 0u || (unsigned)dag->m < k + 64u) return GDN_OK;
  const unsigned base = (unsigned)dag->m - k, kw = k / 64u;
  GDN_TRY(p.core_adj.alloc((size_t)k * kw));
  GDN_TRY(p.core_items_a.alloc((size_t)dag->m));
  GDN_TRY(p.core_ctl.alloc(TC_CORE_CTL_WORDS));
  GDN_TRY(p.core_total.alloc(1));
  GDN_HIP(hipMemsetAsync(p.core_adj.p, 0, (size_t)k * kw * 8, 0));
  GDN_HIP(hipMemsetAsync(p.core_ctl.p, 0, 32, 0));
  hipLaunchKernelGGL(tc_core_adj_kernel, dim3(k / GDN_WAVES_PER_BLOCK), dim3(GDN_BLOCK), 0, 0, dag->rowptr, dag->colidx, base, k, kw,
                     p.core_adj.p);
  hipLaunchKernelGGL(tc_core_items_kernel, dim3(gdn_nblocks((uint64_t)dag->m)), dim3(GDN_BLOCK), 0, 0, dag->rowptr, dag->colidx, dag->m,
                     base, p.core_items_a.p, p.core_ctl.p, p.core_ctl.p + 5);
  GDN_HIP(hipGetLastError());
  unsigned n_items = 0;
  GDN_HIP(hipMemcpy(&n_items, p.core_ctl.p, 4, hipMemcpyDeviceToHost));
  // The core pays where the graph is skewed: symmetrized R-MAT, a third of the rows reach the top ranks twice.  Where hardly a
  // row does (lattice, uniform random, small-world at 2^22-2^24 vertices: its launch and empty grabs cost 0.35 ms of a 3-12 ms
  // count, sessions/r04_98.sh) the hash-set kernel keeps every row.  GDN_TC_CORE set: the caller's choice, whatever the graph.
  if (n_items == 0 || (!gdn_option("GDN_TC_CORE") && n_items < (unsigned)dag->m / 64u)) {
    p.core_adj.release();
    p.core_items_a.release();
    return GDN_OK;
  }
  GDN_TRY(p.core_items_b.alloc((size_t)n_items));
  GDN_TRY(gdn_radix_sort_u64(p.core_items_a.p, p.core_items_b.p, n_items, 62u, 64u, &p.core_items));  // by class, stable
  if (p.core_items == p.core_items_a.p) p.core_items_b.release();
  else p.core_items_a.release();
  {  // GDN_TC_CORE_ASYNC=0 (A/B knob): the core kernel behind tc_count_kernel on the null stream instead of beside it
    const char *e = gdn_test_option("GDN_TC_CORE_ASYNC");
    if (!(e && e[0] == '0')) {  // lowest priority: its workgroups take the wave slots tc_count_kernel's LDS budget leaves free
      int lo_pri = 0, hi_pri = 0;
      if (hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri) != hipSuccess ||
          hipStreamCreateWithPriority(&p.core_stream, hipStreamNonBlocking, lo_pri) != hipSuccess ||
          hipEventCreateWithFlags(&p.core_ready, hipEventDisableTiming) != hipSuccess) {
        // no second stream to be had: the same count, the core kernel behind the hash-set kernel (a full grid)
        (void)hipGetLastError();
        if (p.core_stream) (void)hipStreamDestroy(p.core_stream);
        p.core_stream = nullptr;
        p.core_ready = nullptr;
      }
    }
  }
  p.core_k = k;
  {  // the heavy rows' items of the hash-set kernel now (not by the first count: see tc_core_launch on who starts first)
    unsigned light = 128u;  // (tc_count_rows beside a core)
    if (const char *e = gdn_xoption("GDN_TC_LIGHT")) light = std::max((unsigned)atoi(e), (unsigned)TC_LIGHT_MIN);
    const uint64_t cap64 = dag->nnz / TC_LIGHT_MIN + 1024;
    unsigned long long *it = nullptr;
    unsigned *ct = nullptr;
    GDN_TRY(tc_plan_heavy_items(p, dag, p.dag_in ? p.dag_in : dag, 0, dag->m - (int32_t)k, light,
                                (unsigned)(cap64 > 0x7FFFFFFFull ? 0x7FFFFFFFull : cap64), &it, &ct));
  }
  return GDN_OK;
}
static int tc_core_prepare(gdn_tc_plan &p) {
  GDN_HIP(hipMemsetAsync(p.core_ctl.p + TC_CORE_CUR, 0, (TC_CORE_CTL_WORDS - TC_CORE_CUR) * sizeof(unsigned), 0));
  GDN_HIP(hipMemsetAsync(p.core_total.p, 0, 8, 0));
  if (p.core_stream) {
    GDN_HIP(hipEventRecord(p.core_ready, 0));
    GDN_HIP(hipStreamWaitEvent(p.core_stream, p.core_ready, 0));
  }
  return GDN_OK;
}
static int tc_core_launch(gdn_tc_plan &p, bool tail) {
  const gdn_graph *dag = p.dag;
  const unsigned base = (unsigned)dag->m - p.core_k;
  if (tail && !p.core_stream) return GDN_OK;  // GDN_TC_CORE_ASYNC=0: the one grid already runs behind tc_count_kernel
  // A few workgroups per CU: waves of <= 56 registers and 2-8 KB of LDS each beside tc_count_kernel's four workgroups (133 KB
  // of LDS).  Which of the two kernels reaches a CU first decides how they share it -- a grid that can fill the machine does
  // so when it starts first, and the hash-set kernel then waits for its END (round 4: 31 instead of 23 ms in about half of the
  // runs).  Since round 6 the hash-set kernel is queued FIRST (its heavy-row items are the plan's, nothing runs in front of it).
  // A TAIL grid (on the null stream behind tc_count_kernel, same work list) can finish what is left when the hash-set kernel
  // is done with the whole device -- see below.
  int cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  // (rounds 4-5, under the single work counters: 2 / 3 / 4 at RMAT-23 20.9 / 23.4 / 25.5 ms.  Round 6, striped counters, the hash-set
  // kernel compiled for five waves per SIMD and starting first (its items come from the plan): 2 / 3 / 4 -> RMAT-23 11.65 / 11.18 /
  // 10.85 ms, RMAT-21 2.24 / 2.13 / 2.11, RMAT-24 28.9 / 28.7 / 28.5, Orkut-like 6.84 / 6.83 / 6.81: four)
  unsigned per_cu = 4;
  if (const char *e = gdn_test_option("GDN_TC_CORE_WGS")) per_cu = atoi(e) > 0 ? (unsigned)atoi(e) : per_cu;  // (tuning knob)
  if (!p.core_stream) per_cu = 8;  // GDN_TC_CORE_ASYNC=0: alone on the device
  if (tail) {
    // OFF by default (GDN_TC_CORE_TAIL = workgroups per CU): at RMAT-23 the core's grid is done (18.6 ms) before the hash-set
    // kernel beside it (20.3 ms; alone 15.5), and a tail grid that finds an empty list still costs 0.34 ms behind it
    // (profiles/sessions/r04_103.sh)
    per_cu = 0;
    if (const char *e = gdn_test_option("GDN_TC_CORE_TAIL")) per_cu = (unsigned)atoi(e);
    if (per_cu == 0) return GDN_OK;
  }
  hipStream_t stream = tail ? nullptr : p.core_stream;
  const dim3 grid((unsigned)cus * per_cu), block(GDN_BLOCK);
  unsigned small = TC_CORE_SMALL;  // GDN_TC_CORE_SMALL (tuning knob, 2..64): core lists shorter than this take the pair tests
  if (const char *e = gdn_test_option("GDN_TC_CORE_SMALL")) small = atoi(e) < 2 ? 2u : atoi(e) > 64 ? 64u : (unsigned)atoi(e);
#define TC_CORE_LAUNCH(R)                                                                                                        \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(tc_core_count_kernel<R>), grid, block, 0, stream, dag->rowptr, dag->colidx, p.core_items, \
                     p.core_ctl.p, p.core_adj.p, base, p.core_total.p, small)
  if (p.core_k == 4096u) TC_CORE_LAUNCH(1);
  else if (p.core_k == 8192u) TC_CORE_LAUNCH(2);
  else if (p.core_k == 12288u) TC_CORE_LAUNCH(3);
  else TC_CORE_LAUNCH(4);
#undef TC_CORE_LAUNCH
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

static int tc_copy_graph(const gdn_graph *g, gdn_graph **out) {  // an owned copy (the plan outlives the caller's handle)
  gdn_graph *d = new gdn_graph();
  d->m = g->m;
  d->nnz = g->nnz;
  d->owned = true;
  hipError_t e = gdn_plain_malloc((void **)&d->rowptr, ((size_t)g->m + 1) * sizeof(eoff_t));
  if (e == hipSuccess) e = gdn_plain_malloc((void **)&d->colidx, (g->nnz ? g->nnz : 1) * sizeof(vid_t));
  if (e == hipSuccess) e = hipMemcpy(d->rowptr, g->rowptr, ((size_t)g->m + 1) * sizeof(eoff_t), hipMemcpyDeviceToDevice);
  if (e == hipSuccess && g->nnz) e = hipMemcpy(d->colidx, g->colidx, g->nnz * sizeof(vid_t), hipMemcpyDeviceToDevice);
  if (e != hipSuccess) {
    gdn_set_error("gdn_tc_plan_create: %s", hipGetErrorString(e));
    gdn_graph_free(d);
    return GDN_ERR_OOM;
  }
  *out = d;
  return GDN_OK;
}

int gdn_tc_plan_create(const gdn_graph *g, int32_t oriented, gdn_tc_plan **plan) {
  GDN_REQUIRE(g != nullptr && plan != nullptr, "graph / plan");
  *plan = nullptr;
  HostTimer tprep;
  tprep.start();
  // GDN_TC_FORM: f the forward count on the rank-ordered DAG (default from 2^24 DAG edges on: RMAT-23 38 ms against 66,
  // RMAT-21 8.6 against 9.8, RMAT-19 2.0 against 1.9 -- profiles/r03_tc_forward_ab.txt; its preparation re-ranks and
  // rebuilds the DAG); a the hash-set count on the reference's orientation, u- or v-centric, whichever probes less (default
  // below that); u / v one of the two; bs the wave-per-edge binary-search intersect
  const char *e = gdn_option("GDN_TC_FORM");
  const uint64_t dag_edges = oriented ? g->nnz : g->nnz / 2;
  // (round 6, on the striped work counters, forward + core / the reference's orientation: RMAT-18 0.65 / 0.73 ms, RMAT-19 0.92 / 1.44,
  // RMAT-20 1.29 / 3.10 -- profiles/r06_tc_counters.md; the forward count from 2^22 DAG edges on, 2^24 before)
  const char form = e ? e[0] : (dag_edges >= (1ull << 22) ? 'f' : 'a');
  gdn_tc_plan *p = new gdn_tc_plan();
  int rc = GDN_OK;
  bool forward = form == 'f';
  if (forward) {
    rc = tc_forward_build(g, oriented != 0, &p->dag, &p->dag_in, p->nstart, p->nbound);
    p->form = 3;
    if (rc == 1) {  // an oriented input that is not the reference's orientation of a simple graph: no re-ranking
      forward = false;
      rc = GDN_OK;
    }
  }
  if (!forward) {
    rc = oriented ? tc_copy_graph(g, &p->dag) : tc_orient(g, &p->dag);
    if (rc == GDN_OK) {
      // which formulation probes less (tc_count_kernel): the v-centric one needs the transposed DAG
      unsigned long long probes[2] = {0, 0};
      const bool bs = form == 'b';
      rc = tc_probe_counts(p->dag, probes);
      const bool vform = form == 'v' || (form != 'u' && !bs && (double)probes[1] < 0.85 * (double)probes[0]);
      if (rc == GDN_OK && vform && p->dag->nnz) rc = gdn_graph_transpose(p->dag, &p->dag_in);
      p->form = bs ? 2 : p->dag_in ? 1 : 0;
    }
  }
  if (rc == GDN_OK && forward && p->dag_in) rc = tc_core_build(*p);
  if (rc != GDN_OK) {
    delete p;
    return rc;
  }
  p->prep_ms = tprep.stop_ms();
  *plan = p;
  return GDN_OK;
}

int gdn_tc_plan_count(gdn_tc_plan *plan, uint64_t *total, gdn_stats *stats) {
  GDN_REQUIRE(plan != nullptr && total != nullptr, "plan / total");
  std::lock_guard<std::mutex> one_count(plan->count_mu);  // two threads counting on one plan take turns
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  int rc = GDN_OK;
  *total = 0;
  if (plan->form == 3) {
    if (plan->dag_in) {
      HostTimer tall;
      tall.start();
      rc = tc_count_rows(plan->dag, 0, plan->dag->m - (int32_t)plan->core_k, total, st, plan->dag_in, false, plan->nstart.p,
                         plan->core_k ? plan : nullptr, plan->nbound.p);
      if (plan->core_k && hipStreamSynchronize(plan->core_stream) != hipSuccess && rc == GDN_OK) {
        gdn_set_error("gdn_tc: core count kernel failed: %s", hipGetErrorString(hipGetLastError()));
        rc = GDN_ERR_HIP;
      }
      if (rc == GDN_OK && plan->core_k) {
        unsigned long long hc = 0;
        if (hipMemcpy(&hc, plan->core_total.p, 8, hipMemcpyDeviceToHost) != hipSuccess) {
          gdn_set_error("gdn_tc: core count kernel failed: %s", hipGetErrorString(hipGetLastError()));
          rc = GDN_ERR_HIP;
        }
        *total += hc;
        st.solve_ms = tall.stop_ms() - st.prep_ms;  // (prep_ms so far: the allocations of tc_count_rows)
      }
    }
  } else {
    rc = tc_count_rows(plan->dag, 0, plan->dag->m, total, st, plan->dag_in, plan->form == 2);
  }
  st.prep_ms += plan->prep_ms;
  st.edges_traversed = plan->dag->nnz;  // TEPS = DAG edges / s, src/tc/gpu_base.cu:60
  st.reserved = plan->form | (int)(plan->core_k << 8);  // (bits 8..: ranks of the forward count's core)
  if (stats) *stats = st;
  return rc;
}

int gdn_tc_plan_free(gdn_tc_plan *plan) {
  delete plan;
  return GDN_OK;
}

int gdn_tc_plan_walked_elements(const gdn_tc_plan *plan, uint64_t *elements) {
  GDN_REQUIRE(plan != nullptr && elements != nullptr, "plan / elements");
  *elements = 0;
  if (plan->form != 3 || !plan->dag) return GDN_OK;  // (the other formulations: gdn_tc_probe_counts)
  DevBuf<unsigned long long> acc;
  GDN_TRY(acc.alloc(1));
  GDN_HIP(hipMemset(acc.p, 0, 8));
  const unsigned base = (unsigned)plan->dag->m - plan->core_k;
  hipLaunchKernelGGL(tc_walked_elements_kernel, dim3(4096), dim3(GDN_BLOCK), 0, 0, plan->dag->rowptr, plan->dag->colidx, plan->dag->m, base,
                     acc.p);
  unsigned long long h = 0;
  GDN_HIP(hipMemcpy(&h, acc.p, 8, hipMemcpyDeviceToHost));
  *elements = h;
  return GDN_OK;
}

int gdn_tc_dev(const gdn_graph *g, int32_t oriented, uint64_t *total, gdn_stats *stats) {
  GDN_REQUIRE(g != nullptr && total != nullptr, "graph / total");
  gdn_tc_plan *plan = nullptr;
  GDN_TRY(gdn_tc_plan_create(g, oriented, &plan));
  const int rc = gdn_tc_plan_count(plan, total, stats);
  gdn_tc_plan_free(plan);
  return rc;
}

// The DAG orientation alone (src/common/graph.cc:67-113): what `Graph g(prefix, USE_DAG)` hands to TCSolver.
int gdn_graph_orient(const gdn_graph *g, gdn_graph **dag) {
  GDN_REQUIRE(g != nullptr && dag != nullptr, "graph / dag");
  *dag = nullptr;
  return tc_orient(g, dag);
}

// Triangles whose lowest-ranked vertex lies in the row range [row_lo, row_hi) of an ORIENTED graph: the shard of a
// multi-GPU count (every rank holds the DAG, the ranges partition its rows, the partial counts add up).
int gdn_tc_rows_dev(const gdn_graph *dag, int32_t row_lo, int32_t row_hi, uint64_t *total, gdn_stats *stats) {
  GDN_REQUIRE(dag != nullptr && total != nullptr, "graph / total");
  GDN_REQUIRE(row_lo >= 0 && row_lo <= row_hi && row_hi <= dag->m, "row range");
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  const int rc = tc_count_rows(dag, row_lo, row_hi, total, st);
  if (rc == GDN_OK) {
    eoff_t b[2] = {0, 0};
    GDN_HIP(hipMemcpy(&b[0], dag->rowptr + row_lo, sizeof(eoff_t), hipMemcpyDeviceToHost));
    GDN_HIP(hipMemcpy(&b[1], dag->rowptr + row_hi, sizeof(eoff_t), hipMemcpyDeviceToHost));
    st.edges_traversed = b[1] - b[0];  // DAG edges of the range
  }
  if (stats) *stats = st;
  return rc;
}

// Host API: one call == TCSolver(g, total) (src/tc/main.cc:17).
int gdn_tc(int32_t m, uint64_t nnz, const uint64_t *rowptr, const int32_t *colidx, int32_t oriented,
           uint64_t *total, gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && rowptr && total, "null argument");
  GDN_TRY(gdn_require_device());
  HostTimer th2d;
  th2d.start();
  gdn_graph *g = nullptr;
  GDN_TRY(gdn_graph_upload(m, nnz, rowptr, colidx, &g));
  const double h2d = th2d.stop_ms();
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  int rc = gdn_tc_dev(g, oriented, total, &st);
  st.h2d_ms = h2d;
  gdn_graph_free(g);
  if (stats) *stats = st;
  return rc;
}

}  // extern "C"
