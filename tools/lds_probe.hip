// lds_probe.hip -- measurement tool: throughput of random LDS accesses on gfx950, the primitive
// the propagation-blocked PageRank (gdn_pb.hpp) is built on.  All indices come from registers
// (xorshift), so nothing but the LDS pipe is exercised.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE, int RANDOM>
__global__ void __launch_bounds__(1024) lds_kernel(float *out, int iters, int nwords) {
  extern __shared__ float s[];
  for (int i = threadIdx.x; i < nwords; i += 1024) s[i] = 0.f;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  float acc = 0.f;
  const unsigned mask = nwords - 1;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      unsigned idx;
      if (RANDOM) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; idx = x & mask; }
      else idx = (threadIdx.x + (it * 8 + k) * 1024) & mask;   // conflict free, distinct addresses
      if (MODE == 0) acc += s[idx];                       // ds_read_b32
      else if (MODE == 1) atomicAdd(&s[idx], 1.0f);       // ds_add_f32
      else if (MODE == 2) atomicAdd((unsigned *)&s[idx], 1u);  // ds_add_u32
      else if (MODE == 3) s[idx] = acc + k;               // ds_write_b32
      else if (MODE == 4) { float t = s[idx]; s[idx] = t + 1.0f; }  // read-modify-write, non atomic
      else if (MODE == 5) atomicAdd((unsigned long long *)&s[idx & ~1u], 1ull);  // ds_add_u64
    }
  }
  __syncthreads();
  if (acc == 123.f || s[threadIdx.x & mask] == 7.7f) out[0] = acc;
}

template <int MODE, int RANDOM>
void run(const char *name, float *out, int nwords) {
  const int iters = 256, blocks = 1024;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto k = lds_kernel<MODE, RANDOM>;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, nwords * 4));
  k<<<blocks, 1024, nwords * 4>>>(out, iters, nwords); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(a)); k<<<blocks, 1024, nwords * 4>>>(out, iters, nwords); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  const double ops = (double)blocks * 1024 * iters * 8;
  printf("%-46s %8.3f ms  %9.1f G ops/s  (%.2f lanes/clk/CU at 2.1 GHz, 256 CUs)\n", name, best, ops / best / 1e6,
         ops / best / 1e6 / 256 / 2.1);
}

int main() {
  float *out; CK(hipMalloc(&out, 64));
  for (int nwords : {32768, 8192}) {
    printf("--- LDS table %d KB, 1024-thread workgroups, %d per CU\n", nwords * 4 / 1024, nwords == 32768 ? 1 : 2);
    run<0, 0>("ds_read_b32  conflict-free", out, nwords);
    run<0, 1>("ds_read_b32  random", out, nwords);
    run<3, 0>("ds_write_b32 conflict-free", out, nwords);
    run<3, 1>("ds_write_b32 random", out, nwords);
    run<1, 0>("ds_add_f32   conflict-free", out, nwords);
    run<1, 1>("ds_add_f32   random", out, nwords);
    run<2, 1>("ds_add_u32   random", out, nwords);
    run<5, 1>("ds_add_u64   random", out, nwords);
    run<4, 1>("read+add+write (non atomic) random", out, nwords);
  }
  return 0;
}
