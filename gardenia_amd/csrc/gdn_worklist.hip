// gdn_worklist.hip -- the worklist push primitives of gdn_common.hpp (gdn_wl_push, gdn_wl_push_staged: the wave64
// ballot + prefix-popcount replacements of Worklist::push / Worklist2::push_1item, include/worklistc.h:44-50, :66-89, and
// their CUB block scan) behind one entry of their own, so that they are tested directly and not only through BFS / SSSP /
// BC: every index i with flags[i] != 0 is appended to the queue.
#include "gardenia_hip.h"
#include "gdn_common.hpp"

__global__ void __launch_bounds__(GDN_BLOCK)
wl_filter_kernel(const int32_t *__restrict__ flags, unsigned n, vid_t *__restrict__ queue, unsigned capacity, unsigned *count,
                 unsigned *overflow) {
  // grid-stride in whole waves: the pushes are convergent
  for (unsigned i0 = (blockIdx.x * GDN_BLOCK + threadIdx.x) - gdn_lane(); i0 < n; i0 += gridDim.x * GDN_BLOCK) {
    const unsigned i = i0 + gdn_lane();
    gdn_wl_push(queue, count, capacity, i < n && flags[i] != 0, (vid_t)i, overflow);
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
wl_filter_staged_kernel(const int32_t *__restrict__ flags, unsigned n, vid_t *__restrict__ queue, unsigned capacity,
                        unsigned *count, unsigned *overflow) {
  __shared__ vid_t s_stage[GDN_WAVES_PER_BLOCK][GDN_WL_STAGE];
  GdnWlStage st;
  st.strip = s_stage[threadIdx.x >> 6];
  st.n = 0;
  for (unsigned i0 = (blockIdx.x * GDN_BLOCK + threadIdx.x) - gdn_lane(); i0 < n; i0 += gridDim.x * GDN_BLOCK) {
    const unsigned i = i0 + gdn_lane();
    gdn_wl_push_staged(st, queue, count, capacity, i < n && flags[i] != 0, (vid_t)i, overflow);
  }
  gdn_wl_flush(st, queue, count, capacity, overflow);
}

extern "C" int gdn_worklist_filter_dev(const int32_t *d_flags, int32_t n, int32_t staged, int32_t *d_queue, uint32_t capacity,
                                       uint32_t *d_count, uint32_t *d_overflow) {
  GDN_REQUIRE(n >= 0 && (n == 0 || d_flags != nullptr) && d_queue != nullptr && d_count != nullptr && d_overflow != nullptr,
              "null argument");
  GDN_TRY(gdn_require_device());
  GDN_HIP(hipMemsetAsync(d_count, 0, sizeof(uint32_t), 0));
  GDN_HIP(hipMemsetAsync(d_overflow, 0, sizeof(uint32_t), 0));
  if (n > 0) {
    const unsigned blocks = std::min(gdn_nblocks((uint64_t)n), 2048u);
    if (staged)
      hipLaunchKernelGGL(wl_filter_staged_kernel, dim3(blocks), dim3(GDN_BLOCK), 0, 0, d_flags, (unsigned)n, d_queue, capacity,
                         d_count, d_overflow);
    else
      hipLaunchKernelGGL(wl_filter_kernel, dim3(blocks), dim3(GDN_BLOCK), 0, 0, d_flags, (unsigned)n, d_queue, capacity, d_count,
                         d_overflow);
    GDN_HIP(hipGetLastError());
  }
  GDN_HIP(hipDeviceSynchronize());
  return GDN_OK;
}
