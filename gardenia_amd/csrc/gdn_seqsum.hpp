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
#include <stdint.h>

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

// x (bit pattern) in units of the ulp of a running sum with biased exponent E (1..254): q and whether x sits exactly halfway
GDN_HD void seq_quant(uint32_t xb, uint32_t E, uint32_t &q, uint32_t &tie) {
  const uint32_t ex = xb >> 23;  // sign and exponent
  const uint32_t exn = ex ? ex : 1u;
  const uint32_t mx = (xb & 0x7FFFFFu) | (ex ? 0x800000u : 0u);
  const int d = (int)E - (int)exn;
  tie = 0u;
  if (ex >= 255u) {  // negative (or -0), inf, nan: the hardware adds it
    q = SEQ_CLAMP;
  } else if (d <= 0) {
    const unsigned long long v = (unsigned long long)mx << (-d > 30 ? 30 : -d);
    q = v >= SEQ_CLAMP ? SEQ_CLAMP : (uint32_t)v;
  } else if (d >= 25) {
    q = 0u;
  } else {
    const uint32_t rem = mx & ((1u << d) - 1u), half = 1u << (d - 1);
    q = (mx >> d) + (rem > half ? 1u : 0u);
    tie = rem == half ? 1u : 0u;
  }
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
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {  // inclusive scan under composition
        SeqPair f;
        f.a0 = (uint32_t)__shfl_up((int)p.a0, o, 64);
        f.a1 = (uint32_t)__shfl_up((int)p.a1, o, 64);
        if (lane >= (unsigned)o) p = seq_compose(f, p);
      }
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
