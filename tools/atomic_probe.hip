// atomic_probe.hip -- how many device-scope atomic adds per second ONE address serves, and what spreading them buys.
// Every wave of a machine-filling grid issues `n` adds (lane 0 only, one per wave instruction) to counter (wave mod naddr), the
// counters `stride` bytes apart; returning adds are dependent (the next is issued when the last came back, as a work-item grab
// is), non-returning ones are fire-and-forget (as an epilogue's statistics are).  Round 6: tc_count_kernel handed its work items
// out by one counter and spent 8 of its 15.5 ms there (profiles/r06_tc_counters.md).
// build: make -C tools atomic_probe ; run: tools/_bin/atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void __launch_bounds__(256) probe_kernel(unsigned *ctr, unsigned naddr, unsigned stride_words, int n, int returning,
                                                    unsigned *sink) {
  const unsigned wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  unsigned *p = ctr + (size_t)(wave % naddr) * stride_words;
  unsigned acc = 0;
  if ((threadIdx.x & 63u) == 0u) {
    if (returning) {
      for (int i = 0; i < n; i++) acc += atomicAdd(p + (acc & 0u), 1u);
    } else {
      for (int i = 0; i < n; i++) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (acc == 0xFFFFFFFFu) *sink = acc;
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  unsigned *ctr, *sink;
  const size_t bytes = (size_t)4096 * 4096;
  hipMalloc(&ctr, bytes);
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = cus * 4;  // 16 waves per CU
  printf("%d workgroups of 4 waves, lane 0 of every wave adds\n", blocks);
  for (int returning : {1, 0}) {
    struct { unsigned naddr, stride; } cases[] = {{1, 4}, {2, 4}, {4, 4}, {16, 4}, {4, 128}, {16, 128}, {64, 128}, {256, 128}, {64, 4096}, {1024, 128}};
    for (auto c : cases) {
      const int n = returning ? 64 : 256;
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        hipMemset(ctr, 0, bytes);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(256), 0, 0, ctr, c.naddr, c.stride / 4u, n, returning, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      const double total = (double)blocks * 4 * n;
      printf("%-13s %5u counters %5u bytes apart: %8.1f M adds/s  (%.1f ns per add, %.3f ms for %.0f)\n",
             returning ? "returning" : "non-returning", c.naddr, c.stride, total / best / 1e3, best * 1e6 / total, best, total);
    }
  }
  return 0;
}
