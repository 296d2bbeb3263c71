// gdn_sort.hip -- stable LSD radix sort of 64-bit keys on a bit range: the one device-wide primitive the graph and layout
// builds need (gdn_build.hip: (row << 32 | col) keys of an edge list, the chunk field of the propagation-blocked keys).
// The reference gets its sorts and scans from CUB (include/worklistc.h:6, src/pr/push_pb.cu); this is the wave64 form,
// no library behind it.
//
// One pass = one digit of up to 8 bits:
//   rs_hist_kernel     a workgroup counts the digits of its RS_SPAN keys; counts[digit * nblocks + block]
//   (exclusive scan of that array, digit-major = the global offset of every (digit, block) run)
//   rs_scatter_kernel  the workgroup walks its span tile by tile (RS_TILE keys): every key gets its STABLE rank inside
//                      the tile among the keys of its digit -- per wave step a ballot match (the lanes with my digit), the
//                      earlier waves' counts of the step and a running per-digit count of the tile from LDS --, the tile is
//                      put in digit order in LDS, and written out as runs: consecutive keys of one digit are consecutive
//                      in global memory (16 keys = 128 bytes on average for random digits; a key-at-a-time scatter would
//                      write 8 of every 64 bytes it touches).
// Per pass and key: 8 B read twice, 8 B written.  Wave-level: no lane adds to a shared counter on behalf of another
// lane's digit -- the match leader carries its group's count -- so skewed digits (R-MAT's top row bits) cost no conflicts.
#include "gardenia_hip.h"
#include "gdn_common.hpp"

#define RS_THREADS 256
#define RS_ITEMS 16
#define RS_TILE (RS_THREADS * RS_ITEMS)  // 4096 keys = 32 KB of LDS
#define RS_TILES_PER_SPAN 16
#define RS_SPAN ((unsigned long long)RS_TILE * RS_TILES_PER_SPAN)
#define RS_WAVES (RS_THREADS / 64)

// lanes of the wave whose (valid) digit equals mine; `valid` lanes only
__device__ __forceinline__ unsigned long long rs_match(unsigned d, bool valid, int bits) {
  unsigned long long peers = __ballot(valid);
  for (int b = 0; b < bits; b++) {
    const bool one = (d >> b) & 1u;
    const unsigned long long m = __ballot(one && valid);
    peers &= one ? m : ~m;
  }
  return peers;
}

__global__ void __launch_bounds__(RS_THREADS)
rs_hist_kernel(const unsigned long long *__restrict__ in, unsigned long long n, int shift, int bits, unsigned nblocks,
               uint32_t *__restrict__ counts) {
  __shared__ unsigned s_h[256];
  s_h[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned mask = (1u << bits) - 1u;
  const unsigned long long lo = (unsigned long long)blockIdx.x * RS_SPAN;
  const unsigned long long hi = lo + RS_SPAN < n ? lo + RS_SPAN : n;
  const unsigned long long lt = gdn_lanemask_lt();
  for (unsigned long long i0 = lo + threadIdx.x - gdn_lane(); i0 < hi; i0 += RS_THREADS * 4) {
    unsigned long long k[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const unsigned long long i = i0 + (unsigned long long)r * RS_THREADS + gdn_lane();
      k[r] = i < hi ? __builtin_nontemporal_load(in + i) : 0ull;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const bool valid = i0 + (unsigned long long)r * RS_THREADS + gdn_lane() < hi;
      const unsigned d = (unsigned)(k[r] >> shift) & mask;
      const unsigned long long peers = rs_match(d, valid, bits);
      if (valid && (peers & lt) == 0ull) atomicAdd(&s_h[d], (unsigned)__popcll(peers));
    }
  }
  __syncthreads();
  if (threadIdx.x <= mask) counts[(size_t)threadIdx.x * nblocks + blockIdx.x] = s_h[threadIdx.x];
}

__global__ void __launch_bounds__(RS_THREADS)
rs_scatter_kernel(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out, unsigned long long n, int shift,
                  int bits, unsigned nblocks, const eoff_t *__restrict__ offsets) {
  __shared__ unsigned long long s_keys[RS_TILE];
  __shared__ unsigned long long s_goff[256];  // where the next key of a digit from this workgroup goes
  __shared__ unsigned s_wd[RS_WAVES][256];    // this step: count of a digit in every wave
  __shared__ unsigned s_run[256];             // keys of a digit in the steps of this tile so far
  __shared__ unsigned s_start[256];           // first position of a digit in the tile's digit order
  __shared__ unsigned s_scan[GDN_WAVES_PER_BLOCK];
  const unsigned mask = (1u << bits) - 1u, lane = gdn_lane(), w = threadIdx.x >> 6;
  const unsigned long long lt = gdn_lanemask_lt();
  s_goff[threadIdx.x] = threadIdx.x <= mask ? offsets[(size_t)threadIdx.x * nblocks + blockIdx.x] : 0ull;
  const unsigned long long lo = (unsigned long long)blockIdx.x * RS_SPAN;
  const unsigned long long hi = lo + RS_SPAN < n ? lo + RS_SPAN : n;
  for (unsigned long long base = lo; base < hi; base += RS_TILE) {
    const unsigned cnt = (unsigned)(hi - base < (unsigned long long)RS_TILE ? hi - base : (unsigned long long)RS_TILE);
    s_run[threadIdx.x] = 0u;
#pragma unroll
    for (int ww = 0; ww < RS_WAVES; ww++) s_wd[ww][threadIdx.x] = 0u;
    unsigned long long key[RS_ITEMS];
    unsigned pos[RS_ITEMS];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
      const unsigned j = (unsigned)i * RS_THREADS + threadIdx.x;  // step i holds the tile's keys i*256 .. i*256+255, in order
      key[i] = j < cnt ? __builtin_nontemporal_load(in + base + j) : 0ull;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
      const bool valid = (unsigned)i * RS_THREADS + threadIdx.x < cnt;
      const unsigned d = (unsigned)(key[i] >> shift) & mask;
      const unsigned long long peers = rs_match(d, valid, bits);
      const unsigned rank = (unsigned)__popcll(peers & lt);
      if (valid && rank == 0u) s_wd[w][d] = (unsigned)__popcll(peers);
      __syncthreads();
      if (valid) {
        unsigned o = s_run[d] + rank;
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ww++)
          if ((unsigned)ww < w) o += s_wd[ww][d];
        pos[i] = o;
      }
      __syncthreads();
      {  // thread t keeps digit t's books
        unsigned tot = 0;
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ww++) {
          tot += s_wd[ww][threadIdx.x];
          s_wd[ww][threadIdx.x] = 0u;
        }
        s_run[threadIdx.x] += tot;
      }
      __syncthreads();
    }
    // the tile in digit order
    unsigned total;
    const unsigned mine = s_run[threadIdx.x];
    s_start[threadIdx.x] = gdn_block_excl_scan(mine, s_scan, &total);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
      if ((unsigned)i * RS_THREADS + threadIdx.x < cnt) {
        const unsigned d = (unsigned)(key[i] >> shift) & mask;
        s_keys[s_start[d] + pos[i]] = key[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
      const unsigned j = (unsigned)i * RS_THREADS + threadIdx.x;
      if (j < cnt) {
        const unsigned long long k = s_keys[j];
        const unsigned d = (unsigned)(k >> shift) & mask;
        out[s_goff[d] + (j - s_start[d])] = k;
      }
    }
    __syncthreads();
    s_goff[threadIdx.x] += mine;
    (void)lane;
  }
}

// Sorts the keys by their bits [begin_bit, end_bit), stable.  a holds the input; the result is in *sorted (a or b).
int gdn_radix_sort_u64(unsigned long long *a, unsigned long long *b, unsigned long long n, unsigned begin_bit, unsigned end_bit,
                       const unsigned long long **sorted) {
  *sorted = a;
  if (n < 2 || end_bit <= begin_bit) return GDN_OK;
  if (end_bit > 64) end_bit = 64;
  const unsigned long long nb64 = (n + RS_SPAN - 1) / RS_SPAN;
  GDN_REQUIRE(nb64 < (1ull << 31), "radix sort: key count");
  const unsigned nblocks = (unsigned)nb64;
  // digits as even as 8 bits allow: 12 bits are two passes of 6, not 8 + 4
  const unsigned span = end_bit - begin_bit, passes = (span + 7) / 8;
  DevBuf<uint32_t> counts;
  DevBuf<eoff_t> offsets;
  GDN_TRY(counts.alloc_scratch((size_t)256 * nblocks));
  GDN_TRY(offsets.alloc_scratch((size_t)256 * nblocks + 1));
  unsigned long long *src = a, *dst = b;
  unsigned bit = begin_bit;
  for (unsigned p = 0; p < passes; p++) {
    const int bits = (int)((span - (bit - begin_bit) + (passes - p) - 1) / (passes - p));
    hipLaunchKernelGGL(rs_hist_kernel, dim3(nblocks), dim3(RS_THREADS), 0, 0, src, n, (int)bit, bits, nblocks, counts.p);
    GDN_HIP(hipGetLastError());
    GDN_TRY(gdn_exclusive_scan_u32_to_u64(counts.p, offsets.p, ((size_t)1 << bits) * nblocks, 0));
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(nblocks), dim3(RS_THREADS), 0, 0, src, dst, n, (int)bit, bits, nblocks, offsets.p);
    GDN_HIP(hipGetLastError());
    unsigned long long *t = src;
    src = dst;
    dst = t;
    bit += (unsigned)bits;
  }
  GDN_HIP(hipDeviceSynchronize());
  *sorted = src;
  return GDN_OK;
}

extern "C" int gdn_sort_u64_dev(uint64_t *d_keys, uint64_t *d_tmp, uint64_t n, int32_t begin_bit, int32_t end_bit, uint64_t **d_sorted) {
  GDN_REQUIRE(d_sorted != nullptr && (n == 0 || (d_keys != nullptr && d_tmp != nullptr)), "null argument");
  GDN_REQUIRE(begin_bit >= 0 && end_bit >= begin_bit && end_bit <= 64, "bit range");
  GDN_TRY(gdn_require_device());
  const unsigned long long *sorted = nullptr;
  GDN_TRY(gdn_radix_sort_u64(reinterpret_cast<unsigned long long *>(d_keys), reinterpret_cast<unsigned long long *>(d_tmp), n,
                             (unsigned)begin_bit, (unsigned)end_bit, &sorted));
  *d_sorted = reinterpret_cast<uint64_t *>(const_cast<unsigned long long *>(sorted));
  return GDN_OK;
}
