// memset_probe -- does hipMemset[Async] work on INTERIOR pointers of a stream-ordered (pool) allocation?
// (round 4: the layout builder zeroed sub-arrays of one pooled arena with hipMemsetAsync and read stale data)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x)                                                   \
  do {                                                          \
    hipError_t e_ = (x);                                        \
    if (e_ != hipSuccess) {                                     \
      fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                  \
    }                                                           \
  } while (0)

static int check(const char *what, bool pooled, size_t off, size_t len) {
  const size_t bytes = 1 << 20;
  char *p = nullptr;
  if (pooled) CK(hipMallocAsync((void **)&p, bytes, 0));
  else CK(hipMalloc((void **)&p, bytes));
  CK(hipMemsetAsync(p, 0xCD, bytes, 0));
  CK(hipMemsetAsync(p + off, 0, len, 0));
  std::vector<unsigned char> h(bytes);
  CK(hipMemcpy(h.data(), p, bytes, hipMemcpyDeviceToHost));
  size_t zero_inside = 0, zero_outside = 0;
  for (size_t i = 0; i < bytes; i++) {
    const bool in = i >= off && i < off + len;
    if (h[i] == 0) (in ? zero_inside : zero_outside)++;
  }
  printf("%-22s %s offset %7zu len %6zu: %zu of %zu bytes zeroed inside, %zu zeroed OUTSIDE (%p)\n", what, pooled ? "pool  " : "malloc", off,
         len, zero_inside, len, zero_outside, (void *)p);
  if (pooled) CK(hipFreeAsync(p, 0));
  else CK(hipFree(p));
  CK(hipDeviceSynchronize());
  return zero_inside == len && zero_outside == 0 ? 0 : 1;
}

// hipMemsetAsync(interior, 0) followed by a kernel that sets bits in the same words: does the kernel's result survive, i.e. is
// the memset ordered in front of the kernel on the null stream?
__global__ void set_bits(unsigned *w, unsigned n) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicOr(&w[i], 1u << (i & 31u));
}
static int check_order(bool pooled, size_t big, size_t off, size_t len) {
  char *p = nullptr;
  if (pooled) CK(hipMallocAsync((void **)&p, big, 0));
  else CK(hipMalloc((void **)&p, big));
  CK(hipMemsetAsync(p, 0xCD, big, 0));
  CK(hipMemsetAsync(p + off, 0, len, 0));
  const unsigned nw = (unsigned)(len / 4);
  set_bits<<<(nw + 255) / 256, 256, 0, 0>>>(reinterpret_cast<unsigned *>(p + off), nw);
  std::vector<unsigned> h(nw);
  CK(hipMemcpy(h.data(), p + off, len, hipMemcpyDeviceToHost));
  size_t wrong = 0;
  for (unsigned i = 0; i < nw; i++) wrong += h[i] != (1u << (i & 31u));
  printf("memset -> kernel       %s block %9zu offset %8zu len %8zu: %zu of %u words wrong (first: %08x)\n", pooled ? "pool  " : "malloc", big, off, len,
         wrong, nw, nw ? h[0] : 0u);
  if (pooled) CK(hipFreeAsync(p, 0));
  else CK(hipFree(p));
  CK(hipDeviceSynchronize());
  return wrong ? 1 : 0;
}

int main() {
  CK(hipSetDevice(0));
  int bad = 0;
  for (int rep = 0; rep < 3; rep++)
    for (int pooled = 0; pooled < 2; pooled++)
      for (size_t big : {(size_t)76288, (size_t)(64 << 20)})
        for (size_t len : {(size_t)16, (size_t)360, (size_t)16896}) bad += check_order(pooled != 0, big, 36096, len);
  // a first pooled block, so that the next ones are sub-allocations behind it
  void *hold = nullptr;
  CK(hipMallocAsync(&hold, 12345, 0));
  for (int pooled = 0; pooled < 2; pooled++)
    for (size_t off : {(size_t)0, (size_t)256, (size_t)4096, (size_t)36096, (size_t)65536 + 16})
      for (size_t len : {(size_t)16, (size_t)360, (size_t)16896, (size_t)262144}) bad += check("hipMemsetAsync", pooled != 0, off, len);
  CK(hipFreeAsync(hold, 0));
  CK(hipDeviceSynchronize());
  printf("%s\n", bad ? "BROKEN: interior memsets do not land where they should" : "ok");
  return 0;
}
