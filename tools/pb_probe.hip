// pb_probe.hip -- measurement tool: streaming ceilings of the two propagation-blocking phases
// (gdn_pb.hpp) on synthetic data, one variant per line, to see which resource bounds them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define T 1024

__global__ void fill_u16(uint16_t *p, size_t n, unsigned mask) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) { unsigned long long z = i * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32; p[i] = (uint16_t)(z & mask); }
}
__global__ void fill_g(unsigned *g, size_t n, int mode, unsigned run_groups) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    if (mode == 0) g[i] = (unsigned)i;
    else {  // runs of run_groups groups land at pseudo-random run slots (a permutation of runs)
      size_t run = i / run_groups, off = i % run_groups, nruns = n / run_groups;
      size_t pr = (run * 2654435761ull) % nruns;   // odd multiplier: bijection when nruns is a power of two
      g[i] = (unsigned)(pr * run_groups + off);
    }
  }
}

// phase A variants.  FLAGS bit0: LDS gather, bit1: load G, bit2: nt loads, bit3: 16-B U loads (8 edges/lane)
template <int FLAGS>
__global__ void __launch_bounds__(T) expandA(const float *__restrict__ x, const uint16_t *__restrict__ U,
                                              const unsigned *__restrict__ G, float *__restrict__ vals, size_t per_block) {
  extern __shared__ __attribute__((aligned(16))) float s_x[];
  for (unsigned i = threadIdx.x; i < 32768; i += T) s_x[i] = x[(size_t)blockIdx.x * 32768 + i];
  __syncthreads();
  const size_t h0 = (size_t)blockIdx.x * per_block / 4, h1 = h0 + per_block / 4;
  const u16x4 *U4 = (const u16x4 *)U; f32x4 *X4 = (f32x4 *)vals;
  constexpr int UNR = 8;
  for (size_t h = h0 + threadIdx.x; h < h1; h += UNR * T) {
    u16x4 u[UNR]; unsigned d[UNR];
#pragma unroll
    for (int r = 0; r < UNR; r++) { size_t hh = h + (size_t)r * T; if (hh < h1) {
      u[r] = (FLAGS & 4) ? __builtin_nontemporal_load(U4 + hh) : U4[hh];
      d[r] = (FLAGS & 2) ? ((FLAGS & 4) ? __builtin_nontemporal_load(G + (hh >> 1)) : G[hh >> 1]) : (unsigned)(hh >> 1); } }
#pragma unroll
    for (int r = 0; r < UNR; r++) { size_t hh = h + (size_t)r * T; if (hh < h1) {
      f32x4 o;
      if (FLAGS & 1) { o.x = s_x[u[r].x]; o.y = s_x[u[r].y]; o.z = s_x[u[r].z]; o.w = s_x[u[r].w]; }
      else { o.x = u[r].x; o.y = u[r].y; o.z = u[r].z; o.w = u[r].w; }
      X4[2 * (size_t)d[r] + (hh & 1)] = o; } }
  }
}

// phase B variants. FLAGS bit0: LDS atomics (u64), bit1: fixed conversion, bit2: float LDS accumulate via u32 atomics
template <int FLAGS>
__global__ void __launch_bounds__(T) accumB(const uint16_t *__restrict__ V, const float *__restrict__ vals, float *__restrict__ out,
                                             size_t per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_acc[];
  for (unsigned i = threadIdx.x; i < 16384; i += T) s_acc[i] = 0;
  __syncthreads();
  const size_t q0 = (size_t)blockIdx.x * per_block / 4, q1 = q0 + per_block / 4;
  const f32x4 *X4 = (const f32x4 *)vals; const u16x4 *V4 = (const u16x4 *)V;
  constexpr int UNR = 4;
  float racc = 0.f;
  for (size_t q = q0 + threadIdx.x; q < q1; q += UNR * T) {
    f32x4 xs[UNR]; u16x4 vs[UNR];
#pragma unroll
    for (int r = 0; r < UNR; r++) { size_t qq = q + (size_t)r * T; if (qq < q1) { xs[r] = __builtin_nontemporal_load(X4 + qq); vs[r] = __builtin_nontemporal_load(V4 + qq); } }
#pragma unroll
    for (int r = 0; r < UNR; r++) { size_t qq = q + (size_t)r * T; if (qq < q1) {
      if (FLAGS & 1) {
        unsigned long long a, b, c, d;
        if (FLAGS & 2) { a = (unsigned long long)(xs[r].x * 4.6e18f); b = (unsigned long long)(xs[r].y * 4.6e18f); c = (unsigned long long)(xs[r].z * 4.6e18f); d = (unsigned long long)(xs[r].w * 4.6e18f); }
        else { a = __float_as_uint(xs[r].x); b = __float_as_uint(xs[r].y); c = __float_as_uint(xs[r].z); d = __float_as_uint(xs[r].w); }
        atomicAdd(&s_acc[vs[r].x & 16383], a); atomicAdd(&s_acc[vs[r].y & 16383], b); atomicAdd(&s_acc[vs[r].z & 16383], c); atomicAdd(&s_acc[vs[r].w & 16383], d);
      } else racc += xs[r].x + xs[r].y + xs[r].z + xs[r].w + vs[r].x + vs[r].y + vs[r].z + vs[r].w;
    } }
  }
  __syncthreads();
  if (racc == 1.2345f || s_acc[threadIdx.x] == 77) out[0] = racc;
}

// ceilings: write-only and copy streams, 16 B per lane
__global__ void __launch_bounds__(T) write_only(f32x4 *__restrict__ out, size_t n4) {
  size_t i = (size_t)blockIdx.x * T * 8 + threadIdx.x;
  f32x4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
  for (int k = 0; k < 8; k++) { size_t j = i + (size_t)k * T; if (j < n4) out[j] = v; }
}
__global__ void __launch_bounds__(T) write_only_nt(f32x4 *__restrict__ out, size_t n4) {
  size_t i = (size_t)blockIdx.x * T * 8 + threadIdx.x;
  f32x4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
  for (int k = 0; k < 8; k++) { size_t j = i + (size_t)k * T; if (j < n4) __builtin_nontemporal_store(v, out + j); }
}
__global__ void __launch_bounds__(T) copy16(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, size_t n4) {
  size_t i = (size_t)blockIdx.x * T * 8 + threadIdx.x;
  f32x4 v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { size_t j = i + (size_t)k * T; if (j < n4) v[k] = __builtin_nontemporal_load(in + j); }
#pragma unroll
  for (int k = 0; k < 8; k++) { size_t j = i + (size_t)k * T; if (j < n4) out[j] = v[k]; }
}

template <class F> float timeit(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 4; r++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
  return best;
}

int main() {
  const size_t nblk = 4096, per_block = (size_t)1 << 19;       // 2^31 edges
  const size_t n = nblk * per_block;
  uint16_t *U, *V; unsigned *G; float *vals, *x, *out;
  CK(hipMalloc(&U, n * 2)); CK(hipMalloc(&V, n * 2)); CK(hipMalloc(&G, n / 8 * 4)); CK(hipMalloc(&vals, n * 4));
  CK(hipMalloc(&x, nblk * 32768 * 4)); CK(hipMalloc(&out, 64));
  CK(hipMemset(x, 0, nblk * 32768 * 4)); CK(hipMemset(vals, 0, n * 4));
  const bool random_data = getenv("PB_PROBE_RANDOM") != nullptr;  // zero-filled vs random payloads
  if (random_data) {
    fill_u16<<<8192, 256>>>((uint16_t *)x, nblk * 32768 * 2, 0x3F7F);   // random small positive floats (bit patterns)
    fill_u16<<<8192, 256>>>((uint16_t *)vals, n * 2, 0x3F7F);
    CK(hipDeviceSynchronize());
  }
  printf("payload: %s\n", random_data ? "random" : "zeros");
  fill_u16<<<8192, 256>>>(U, n, 32767); fill_u16<<<8192, 256>>>(V, n, 16383); CK(hipDeviceSynchronize());
  const int ldsA = 32768 * 4 + 16, ldsB = 16384 * 8;
#define RUNA(FL, name) { auto k = expandA<FL>; CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, ldsA)); \
    float ms = timeit([&] { k<<<nblk, T, ldsA>>>(x, U, G, vals, per_block); }); \
    printf("A %-52s %7.3f ms  %6.2f TB/s (6.5 B/edge)\n", name, ms, n * 6.5 / ms / 1e9); }
#define RUNB(FL, name) { auto k = accumB<FL>; CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, ldsB)); \
    float ms = timeit([&] { k<<<nblk * 2, T, ldsB>>>(V, vals, out, per_block / 2); }); \
    printf("B %-52s %7.3f ms  %6.2f TB/s (6 B/edge)\n", name, ms, n * 6.0 / ms / 1e9); }
  for (int mode = 0; mode < 4; mode++) {
    const unsigned run_groups = mode == 1 ? 8 : mode == 2 ? 16 : 32;
    fill_g<<<8192, 256>>>(G, n / 8, mode ? 1 : 0, run_groups); CK(hipDeviceSynchronize());
    printf("--- G = %s\n", mode == 0 ? "identity (sequential stores)" : mode == 1 ? "runs of 64 edges (256 B) permuted" : mode == 2 ? "runs of 128 edges (512 B) permuted" : "runs of 256 edges (1 KB) permuted");
    RUNA(7, "full: LDS gather + G + nt loads");
    if (mode == 0) {
      RUNA(6, "no LDS gather");
      RUNA(5, "no G load (sequential)");
      RUNA(3, "plain (temporal) loads");
      RUNA(4, "no LDS, no G: pure 2B-read/4B-write stream");
    }
  }
  {
    const size_t n4 = n / 4;  // 8.6 GB
    const unsigned nb = (unsigned)((n4 + T * 8 - 1) / (T * 8));
    float ms = timeit([&] { write_only<<<nb, T>>>((f32x4 *)vals, n4); });
    printf("W write-only 16 B/lane (plain)                         %7.3f ms  %6.2f TB/s\n", ms, n * 4.0 / ms / 1e9);
    ms = timeit([&] { write_only_nt<<<nb, T>>>((f32x4 *)vals, n4); });
    printf("W write-only 16 B/lane (nontemporal)                   %7.3f ms  %6.2f TB/s\n", ms, n * 4.0 / ms / 1e9);
    const size_t h4 = n4 / 2;
    const unsigned nb2 = (unsigned)((h4 + T * 8 - 1) / (T * 8));
    ms = timeit([&] { copy16<<<nb2, T>>>((const f32x4 *)vals, (f32x4 *)vals + h4, h4); });
    printf("C copy 4.3 GB -> 4.3 GB                                %7.3f ms  %6.2f TB/s (read+write)\n", ms, n * 4.0 / ms / 1e9);
  }
  RUNB(3, "full: u64 LDS atomics + fixed conversion");
  RUNB(1, "u64 LDS atomics, no conversion");
  RUNB(0, "no atomics: pure 6 B/edge read stream");
  return 0;
}
