// gather_probe.hip -- measurement tool (not part of the product library).
// Prices the primitive every CSR kernel here is made of: N random 4-byte gathers out of a
// table of T bytes, next to a coalesced index stream.  Reports G gathers/s for table sizes
// that sit in L2 (4 MiB/XCD), Infinity Cache (256 MiB) and HBM, for uniform and R-MAT-skewed
// index distributions, plus the plain streaming rate, so DESIGN.md can state which bound the
// PageRank pull kernel runs against.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z;
}

// idx[i] uniform in [0, n)   (mode 0)   or RMAT-like skew: each bit 1 with p=.24 (mode 1),
// mode 2: skewed AND hub-sorted (popular ids are the small ones)
__global__ void make_idx(int *idx, size_t n_idx, unsigned n_table_log2, int mode) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n_idx; i += stride) {
    unsigned long long h = mix64(i * 0x9E3779B97F4A7C15ull + 12345);
    unsigned v = 0;
    if (mode == 0) v = (unsigned)(h >> 11) & ((1u << n_table_log2) - 1);
    else {
      unsigned long long h2 = mix64(h + 77);
      for (unsigned b = 0; b < n_table_log2; b++) {
        unsigned r = (unsigned)((b & 1 ? h : h2) >> ((b >> 1) * 4)) & 0xFF;  // 8-bit draw, reused bits ok for a probe
        r = (unsigned)(mix64(h + b) & 0xFF);
        v = (v << 1) | (r < 61 ? 1u : 0u);  // p(1) = 61/256 = .238
      }
      if (mode == 1) v = (unsigned)(mix64(v * 0x9E3779B1ull) & ((1u << n_table_log2) - 1)) ;  // scatter hubs (not a bijection; fine for a probe)
      else {
        // hub-sorted: rank roughly by popcount -> put low-popcount ids first: reverse bits so
        // the popular (few ones) ids cluster near 0
        v = v;  // ids with few 1 bits are numerically small on average already
      }
    }
    idx[i] = (int)v;
  }
}

__global__ void __launch_bounds__(256) gather_kernel(const int *__restrict__ idx, const float *__restrict__ table,
                                                      size_t n_idx, float *__restrict__ out) {
  const size_t tile = (size_t)blockIdx.x * 4096;
  float acc = 0.f;
  int c[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const size_t j = tile + k * 256 + threadIdx.x;
    c[k] = j < n_idx ? __builtin_nontemporal_load(idx + j) : 0;
  }
#pragma unroll
  for (int k = 0; k < 16; k++) acc += table[c[k]];
  if (acc == 123.456f) out[0] = acc;
}

__global__ void __launch_bounds__(256) stream_kernel(const int *__restrict__ idx, size_t n_idx, float *__restrict__ out) {
  const size_t tile = (size_t)blockIdx.x * 4096;
  int acc = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const size_t j = tile + k * 256 + threadIdx.x;
    acc += j < n_idx ? __builtin_nontemporal_load(idx + j) : 0;
  }
  if (acc == 123456789) out[0] = (float)acc;
}

typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stream16_kernel(const v4i *__restrict__ idx, size_t n4, float *__restrict__ out) {
  const size_t tile = (size_t)blockIdx.x * 1024;
  int acc = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const size_t j = tile + k * 256 + threadIdx.x;
    if (j < n4) { v4i v = __builtin_nontemporal_load(idx + j); acc += v.x + v.y + v.z + v.w; }
  }
  if (acc == 123456789) out[0] = (float)acc;
}

// LDS gather: table of 32K floats in LDS, indices streamed
__global__ void __launch_bounds__(256) lds_gather_kernel(const int *__restrict__ idx, const float *__restrict__ table,
                                                          size_t n_idx, float *__restrict__ out, int tiles_per_block) {
  __shared__ float s_t[32768];
  for (int i = threadIdx.x; i < 32768; i += 256) s_t[i] = table[i];
  __syncthreads();
  float acc = 0.f;
  for (int t = 0; t < tiles_per_block; t++) {
    const size_t tile = ((size_t)blockIdx.x * tiles_per_block + t) * 4096;
    int c[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const size_t j = tile + k * 256 + threadIdx.x;
      c[k] = j < n_idx ? __builtin_nontemporal_load(idx + j) : 0;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) acc += s_t[c[k] & 32767];
  }
  if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char **argv) {
  const size_t n_idx = argc > 1 ? (size_t)atoll(argv[1]) : ((size_t)1 << 29);
  int *idx; float *table, *out;
  CK(hipMalloc(&idx, n_idx * 4));
  CK(hipMalloc(&table, (size_t)1 << 31));  // up to 2 GiB table
  CK(hipMalloc(&out, 64));
  CK(hipMemset(table, 0, (size_t)1 << 31));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const unsigned nblk = (unsigned)((n_idx + 4095) / 4096);
  printf("n_idx = %zu (%.2f GB index stream)\n", n_idx, n_idx * 4 / 1e9);
  auto timeit = [&](auto launch, const char *name, double items) {
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
      CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %8.2f G items/s  %8.2f GB/s (4B/item)\n", name, best, items / best / 1e6, items * 4 / best / 1e6);
  };
  make_idx<<<4096, 256>>>(idx, n_idx, 20, 0); CK(hipDeviceSynchronize());
  timeit([&] { stream_kernel<<<nblk, 256>>>(idx, n_idx, out); }, "stream idx only, dword/lane", (double)n_idx);
  timeit([&] { stream16_kernel<<<(unsigned)((n_idx / 4 + 1023) / 1024), 256>>>((const v4i *)idx, n_idx / 4, out); }, "stream idx only, dwordx4/lane", (double)n_idx);
  for (int mode = 0; mode < 3; mode++) {
    for (unsigned lg : {15u, 18u, 20u, 22u, 24u, 25u, 27u, 29u}) {
      make_idx<<<4096, 256>>>(idx, n_idx, lg, mode); CK(hipDeviceSynchronize());
      char name[128];
      snprintf(name, sizeof(name), "gather mode=%s table=%7.2f MiB", mode == 0 ? "uniform" : mode == 1 ? "skew-scattered" : "skew-hubsorted", (4.0 * (1ull << lg)) / (1 << 20));
      timeit([&] { gather_kernel<<<nblk, 256>>>(idx, table, n_idx, out); }, name, (double)n_idx);
    }
  }
  make_idx<<<4096, 256>>>(idx, n_idx, 15, 0); CK(hipDeviceSynchronize());
  {
    const int tpb = 64;
    const unsigned nb2 = (unsigned)((n_idx + 4096ull * tpb - 1) / (4096ull * tpb));
    timeit([&] { lds_gather_kernel<<<nb2, 256>>>(idx, table, n_idx, out, tpb); }, "LDS gather (32K-entry table, uniform)", (double)n_idx);
  }
  return 0;
}
