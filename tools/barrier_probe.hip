// barrier_probe.hip -- what one gdn_grid_barrier costs on a cooperative grid of one workgroup per CU (the floor of a fused
// light level): N barriers back to back, with and without a device-scope round trip of "work" between them.
// build: make -C tools barrier_probe ; run: tools/_bin/barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../gardenia_amd/csrc/gdn_common.hpp"

__global__ void __launch_bounds__(256) probe_kernel(unsigned *bar, unsigned *scratch, int n, int work) {
  unsigned acc = 0;
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < work; k++)  // dependent device-scope round trips
      acc += atomicAdd(scratch + 64 * ((blockIdx.x * 256 + threadIdx.x + acc) & 1023u), 1u);
    gdn_grid_barrier(bar, gridDim.x);
  }
  if (acc == 0xFFFFFFFFu) scratch[0] = acc;
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  unsigned *bar, *scratch;
  hipMalloc(&bar, 256);
  hipMalloc(&scratch, 64 * 1024 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int blocks : {32, 128, cus}) {
    for (int work : {0, 1, 4}) {
      int n = 2000;
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        hipMemset(bar, 0, 256);
        hipMemset(scratch, 0, 64 * 1024 * 4);
        void *args[] = {&bar, &scratch, &n, &work};
        hipEventRecord(e0, 0);
        if (hipLaunchCooperativeKernel((const void *)probe_kernel, dim3(blocks), dim3(256), args, 0, 0) != hipSuccess) {
          printf("launch failed\n");
          return 1;
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("%3d workgroups, %d dependent atomics between barriers: %.2f us per iteration\n", blocks, work, 1e3f * best / n);
    }
  }
  return 0;
}
