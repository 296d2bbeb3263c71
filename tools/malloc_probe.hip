// malloc_probe -- what a device allocation costs on this box (round 4: the layout build of a one-shot solve is a
// few milliseconds of kernels, so hipMalloc / hipFree / first touch decide its wall time).
//   for each size: fresh hipMalloc, first-touch memset, second memset, hipFree, then the same size again (does the
//   runtime hand the block back without mapping it again?), and the same through the stream-ordered pool
//   (hipMallocAsync with an unlimited release threshold).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_));                \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

int main() {
  CK(hipSetDevice(0));
  CK(hipFree(nullptr));
  const size_t sizes[] = {(size_t)1 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)256 << 20, (size_t)1 << 30, (size_t)4 << 30, (size_t)16 << 30};
  printf("# hipMalloc / hipFree, synchronous\n");
  printf("%12s %10s %10s %10s %10s | %10s %10s %10s\n", "bytes", "malloc", "touch1", "touch2", "free", "malloc_b", "touch1_b", "free_b");
  for (size_t s : sizes) {
    double t[8];
    void *p = nullptr;
    for (int rep = 0; rep < 2; rep++) {
      double t0 = now_ms();
      CK(hipMalloc(&p, s));
      t[rep * 4 + 0] = now_ms() - t0;
      t0 = now_ms();
      CK(hipMemset(p, 1, s));
      CK(hipDeviceSynchronize());
      t[rep * 4 + 1] = now_ms() - t0;
      t0 = now_ms();
      CK(hipMemset(p, 2, s));
      CK(hipDeviceSynchronize());
      t[rep * 4 + 2] = now_ms() - t0;
      t0 = now_ms();
      CK(hipFree(p));
      t[rep * 4 + 3] = now_ms() - t0;
    }
    printf("%12zu %10.3f %10.3f %10.3f %10.3f | %10.3f %10.3f %10.3f\n", s, t[0], t[1], t[2], t[3], t[4], t[5], t[7]);
  }
  printf("# hipMallocAsync on the default pool, release threshold = unlimited\n");
  hipMemPool_t pool;
  CK(hipDeviceGetDefaultMemPool(&pool, 0));
  uint64_t thr = UINT64_MAX;
  CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
  printf("%12s %10s %10s %10s | %10s %10s %10s\n", "bytes", "malloc", "touch1", "free", "malloc_b", "touch1_b", "free_b");
  for (size_t s : sizes) {
    double t[6];
    void *p = nullptr;
    for (int rep = 0; rep < 2; rep++) {
      double t0 = now_ms();
      CK(hipMallocAsync(&p, s, 0));
      CK(hipStreamSynchronize(0));
      t[rep * 3 + 0] = now_ms() - t0;
      t0 = now_ms();
      CK(hipMemsetAsync(p, 1, s, 0));
      CK(hipStreamSynchronize(0));
      t[rep * 3 + 1] = now_ms() - t0;
      t0 = now_ms();
      CK(hipFreeAsync(p, 0));
      CK(hipStreamSynchronize(0));
      t[rep * 3 + 2] = now_ms() - t0;
    }
    printf("%12zu %10.3f %10.3f %10.3f | %10.3f %10.3f %10.3f\n", s, t[0], t[1], t[2], t[3], t[4], t[5]);
  }
  // a smaller block out of a pool that holds a larger freed one
  {
    void *p = nullptr;
    double t0 = now_ms();
    CK(hipMallocAsync(&p, (size_t)3 << 30, 0));
    CK(hipStreamSynchronize(0));
    printf("pool: 3 GiB after the 16 GiB block was freed: %.3f ms\n", now_ms() - t0);
    CK(hipFreeAsync(p, 0));
    CK(hipStreamSynchronize(0));
    size_t used = 0, resv = 0;
    CK(hipMemPoolGetAttribute(pool, hipMemPoolAttrReservedMemCurrent, &resv));
    CK(hipMemPoolGetAttribute(pool, hipMemPoolAttrUsedMemCurrent, &used));
    printf("pool reserved %.2f GiB, used %.2f GiB\n", resv / 1073741824.0, used / 1073741824.0);
  }
  // many small allocations (the layout build makes ~40 of them)
  {
    void *q[64];
    double t0 = now_ms();
    for (int i = 0; i < 64; i++) CK(hipMalloc(&q[i], 4096 + 1024 * i));
    const double a = now_ms() - t0;
    t0 = now_ms();
    for (int i = 0; i < 64; i++) CK(hipFree(q[i]));
    printf("64 small hipMalloc: %.3f ms, 64 hipFree: %.3f ms\n", a, now_ms() - t0);
    t0 = now_ms();
    for (int i = 0; i < 64; i++) CK(hipMallocAsync(&q[i], 4096 + 1024 * i, 0));
    CK(hipStreamSynchronize(0));
    const double b = now_ms() - t0;
    t0 = now_ms();
    for (int i = 0; i < 64; i++) CK(hipFreeAsync(q[i], 0));
    CK(hipStreamSynchronize(0));
    printf("64 small hipMallocAsync: %.3f ms, 64 hipFreeAsync: %.3f ms\n", b, now_ms() - t0);
  }
  // a blocking 8-byte D2H copy and an empty-kernel round trip (the build reads a few counters back)
  {
    void *d = nullptr;
    CK(hipMalloc(&d, 64));
    unsigned long long h = 0;
    CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    double t0 = now_ms();
    for (int i = 0; i < 100; i++) CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("blocking 8-byte D2H: %.1f us each\n", (now_ms() - t0) * 10.0);
    unsigned long long *hp = nullptr;
    CK(hipHostMalloc((void **)&hp, 64, hipHostMallocMapped));
    t0 = now_ms();
    for (int i = 0; i < 100; i++) {
      CK(hipMemcpyAsync(hp, d, 8, hipMemcpyDeviceToHost, 0));
      CK(hipStreamSynchronize(0));
    }
    printf("async 8-byte D2H into pinned memory + sync: %.1f us each\n", (now_ms() - t0) * 10.0);
    CK(hipFree(d));
  }
  return 0;
}
