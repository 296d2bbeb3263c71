// gdn_seqsum.hpp -- the reference's row sum, bit for bit, without its chain of dependent additions.
//
// src/pr/omp_base.cc:27-30 adds the contributions of a row one by one in fp32:  S <- fl(S + x_k), k in CSR order.  The result
// depends on the order, so a row of 1.3 M in-edges (RMAT-27's longest) is a chain of 1.3 M dependent additions -- 5 ms at the
// 8-10 cycles an add through v_readlane costs, longer than a whole iteration.  But the chain has structure:
//
//   While S stays inside one binade [2^e, 2^(e+1)), S is an integer multiple P of u = ulp(S) = 2^(e-23), 2^23 <= P < 2^24,
//   and for x >= 0
//        fl(S + x) = (P + q + t) u,   q = x / u rounded to nearest (ties DOWN),
//                                     t = 1 iff x / u lies exactly halfway AND P + q is odd   (round half to even)
//   -- an INTEGER recurrence whose only dependence on the running value is its parity.  An element is therefore a function on
//   (value, parity) of the form p -> p + a[p mod 2]; such functions are closed under composition ((f;g)[r] = f[r] + g[(r +
//   f[r]) mod 2]) and composition is associative: a wave evaluates 512 elements with one parallel scan of pairs (a[0], a[1]).
//   When the prefix reaches 2^24 the binade ends: the running sum in front of the lane where that happens is exact, that
//   lane's elements are added by the hardware, one by one, and the scan of the remaining lanes is redone on the new binade.
//   A row crosses at most ~40 binades, so all but a few dozen of its additions are settled by scans.
//   Everything that is not a plain non-negative finite addend or a normal running sum (S = 0 at the start, denormals, -0,
//   negative values, inf, nan) takes the hardware path too: the result is the sequential fp32 sum in every case
//   (tests/test_seqsum_math.py emulates exactly this on the CPU against numpy; tests/test_gpu_parity.py runs the kernel).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef GDN_HD
#ifdef __HIPCC__
#define GDN_HD __host__ __device__ __forceinline__
#else
#define GDN_HD inline
#endif
#endif

#define SEQ_CLAMP (1u << 23)  // any q >= 2^23 ends the binade (P >= 2^23); clamping keeps 512-element prefixes below 2^32

struct SeqPair {
  uint32_t a0, a1;  // what the segment adds to an integer prefix of even / odd parity
};

// x (bit pattern) in units of the ulp of a running sum with biased exponent E (1..254): q = x / ulp rounded DOWN to an integer
// plus 1 when the remainder exceeds one half, tie = the remainder IS one half.  A row's blocks are a chain and what a block
// costs is this code, so it is float arithmetic that happens to be exact: y = x * 2^(150 - E) is x scaled by a power of two
// (exact for every finite x: the product of a float and a power of two that neither overflows nor leaves the denormal range;
// denormal RESULTS could round, but y < 1/2 then and the outcome q = 0, no tie is the same), floor(y) and y - floor(y) are exact.
//   y not in [0, 2^23) -- x too large for this binade, negative, inf, nan --: q = SEQ_CLAMP, the hardware adds that element.
// (-0 reads as +0: S + -0 = S for the positive S this is used with.)
GDN_HD void seq_quant(uint32_t xb, uint32_t E, uint32_t &q, uint32_t &tie) {
#ifdef __HIP_DEVICE_COMPILE__
  const float x = __uint_as_float(xb);
  const float y = __builtin_amdgcn_ldexpf(x, 150 - (int)E);  // v_ldexp_f32
  const float fl = __builtin_floorf(y), fr = y - fl;
#else
  float x;
  memcpy(&x, &xb, 4);
  const float y = ldexpf(x, 150 - (int)E);
  const float fl = floorf(y), fr = y - fl;
#endif
  const bool in = y >= 0.0f && y < 8388608.0f;  // (false for nan)
  const uint32_t qi = (uint32_t)(int)fl + (fr > 0.5f ? 1u : 0u);
  q = in ? qi : SEQ_CLAMP;
  tie = (in && fr == 0.5f) ? 1u : 0u;
}

// the segment followed by one element
GDN_HD void seq_push(SeqPair &p, uint32_t q, uint32_t tie) {
  uint32_t t0 = p.a0 + q, t1 = p.a1 + q;
  t0 += tie & t0;         // incoming parity 0: the prefix is odd iff t0 is
  t1 += tie & (t1 + 1u);  // incoming parity 1
  p.a0 = t0;
  p.a1 = t1;
}

// segment f followed by segment g
GDN_HD SeqPair seq_compose(const SeqPair &f, const SeqPair &g) {
  SeqPair h;
  h.a0 = f.a0 + ((f.a0 & 1u) ? g.a1 : g.a0);
  h.a1 = f.a1 + (((f.a1 + 1u) & 1u) ? g.a1 : g.a0);
  return h;
}

#ifdef __HIPCC__
// inclusive scan of the lanes' pairs under composition: inside the rows of 16 lanes by DPP row_shr (a lane without a source reads
// the pair (0, 0) = the identity -- no select), across the four rows through the rows' totals in scalar registers
// ... inside every row of 16 lanes (four DPP steps)
__device__ __forceinline__ SeqPair seq_row_scan(SeqPair p) {
#define SEQ_ROW_STEP(CTRL)                                                                       \
  {                                                                                              \
    SeqPair f;                                                                                   \
    f.a0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p.a0, CTRL, 0xf, 0xf, true);            \
    f.a1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p.a1, CTRL, 0xf, 0xf, true);            \
    p = seq_compose(f, p);                                                                       \
  }
  SEQ_ROW_STEP(0x111)  // row_shr:1
  SEQ_ROW_STEP(0x112)  // row_shr:2
  SEQ_ROW_STEP(0x114)  // row_shr:4
  SEQ_ROW_STEP(0x118)  // row_shr:8
#undef SEQ_ROW_STEP
  return p;
}
__device__ __forceinline__ SeqPair seq_wave_scan(SeqPair p, unsigned lane) {
  p = seq_row_scan(p);
  SeqPair t0, t1, t2;
  t0.a0 = (uint32_t)__builtin_amdgcn_readlane((int)p.a0, 15);
  t0.a1 = (uint32_t)__builtin_amdgcn_readlane((int)p.a1, 15);
  t1.a0 = (uint32_t)__builtin_amdgcn_readlane((int)p.a0, 31);
  t1.a1 = (uint32_t)__builtin_amdgcn_readlane((int)p.a1, 31);
  t2.a0 = (uint32_t)__builtin_amdgcn_readlane((int)p.a0, 47);
  t2.a1 = (uint32_t)__builtin_amdgcn_readlane((int)p.a1, 47);
  const SeqPair t01 = seq_compose(t0, t1), t012 = seq_compose(t01, t2);
  const unsigned row = lane >> 4;
  SeqPair f = {0u, 0u};
  f = row == 1u ? t0 : f;
  f = row == 2u ? t01 : f;
  f = row == 3u ? t012 : f;
  return seq_compose(f, p);
}

// The block as ONE function: what its 64 x N elements add to a prefix of even / odd parity on the binade of exponent E --
// valid as long as the prefix stays below 2^24 (the caller tests that on the total: every term is >= 0).  This is what lets
// several waves work on consecutive blocks of one row at once: each computes its block's pair on the binade the row is in,
// one cheap pass chains the pairs (pr_refseg_wg_kernel).
template <int N>
__device__ __forceinline__ SeqPair seq_block_pair(uint32_t E, const uint32_t (&x)[N], unsigned lane) {
  SeqPair p = {0u, 0u};
#pragma unroll
  for (int k = 0; k < N; k++) {
    uint32_t q, tie;
    seq_quant(x[k], E, q, tie);
    seq_push(p, q, tie);
  }
  p = seq_wave_scan(p, lane);
  SeqPair t;
  t.a0 = (uint32_t)__builtin_amdgcn_readlane((int)p.a0, 63);
  t.a1 = (uint32_t)__builtin_amdgcn_readlane((int)p.a1, 63);
  return t;
}

// One block of 64 x N elements in order (lane l holds elements N l .. N l + N - 1 as bit patterns), running sum S (bit
// pattern, wave-uniform) -> the running sum behind the block.  All 64 lanes call it.
template <int N>
__device__ __forceinline__ uint32_t seq_block(uint32_t S, const uint32_t (&x)[N], unsigned lane) {
  unsigned start = 0;  // lanes below are settled
  for (;;) {
    const uint32_t E = S >> 23;
    unsigned L;
    if (E - 1u < 254u) {  // a normal positive running sum
      SeqPair p = {0u, 0u};
      if (lane >= start) {
#pragma unroll
        for (int k = 0; k < N; k++) {
          uint32_t q, tie;
          seq_quant(x[k], E, q, tie);
          seq_push(p, q, tie);
        }
      }
      p = seq_wave_scan(p, lane);
      const uint32_t P0 = (S & 0x7FFFFFu) | 0x800000u;
      const uint32_t tot = P0 + ((P0 & 1u) ? p.a1 : p.a0);
      const unsigned long long cross = __ballot(lane >= start && tot >= (1u << 24));
      if (!cross) return (E << 23) | ((uint32_t)__builtin_amdgcn_readlane((int)tot, 63) & 0x7FFFFFu);
      L = (unsigned)__ffsll((long long)cross) - 1u;
      const uint32_t Pb = L ? (uint32_t)__builtin_amdgcn_readlane((int)tot, L - 1u) : P0;  // lanes below `start` hold P0
      S = (E << 23) | (Pb & 0x7FFFFFu);
    } else {
      L = start;
    }
    float s = __uint_as_float(S);
#pragma unroll
    for (int k = 0; k < N; k++) s = gdn_fadd(s, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)x[k], L)));
    S = __float_as_uint(s);
    start = L + 1u;
    if (start >= 64u) return S;
  }
}
#endif
