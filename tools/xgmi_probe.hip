// xgmi_probe -- the first thing to run on any lease with >= 2 GPUs: what does the exchange of a
// sharded PageRank iteration cost on THIS node?  One process, one host thread per device.
//   1. peer copies: device 0 -> device d for every d, and all pairs at once (every device sends its slice to every
//      other: the direct all-to-all pattern the fully connected xGMI mesh should serve at ~N-1 links per device);
//   2. ncclAllGather (in place, ncclFloat) of the same totals through RCCL: does it drive the N-1 direct links side by
//      side, or ring through one at a time?  (252 MB = the squished contribution vector of RMAT-27.)
// Prints MB, ms and GB/s per receiving device; nothing here is linked into libgardenia_hip.so.
//   build:  make -C tools xgmi_probe        run:  tools/_bin/xgmi_probe [ndev]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)
#define NK(x)                                                                            \
  do {                                                                                   \
    ncclResult_t r_ = (x);                                                               \
    if (r_ != ncclSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, ncclGetErrorString(r_)); \
      exit(1);                                                                           \
    }                                                                                    \
  } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  int ndev = 0;
  CK(hipGetDeviceCount(&ndev));
  if (argc > 1 && atoi(argv[1]) > 0 && atoi(argv[1]) < ndev) ndev = atoi(argv[1]);
  if (ndev < 2) {
    printf("xgmi_probe: %d device(s) visible -- needs two or more\n", ndev);
    return 0;
  }
  const size_t totals_mb[] = {32, 64, 252};
  std::vector<float *> buf(ndev);
  std::vector<hipStream_t> st(ndev);
  const size_t max_floats = (252ull << 20) / 4;
  for (int d = 0; d < ndev; d++) {
    CK(hipSetDevice(d));
    CK(hipMalloc((void **)&buf[d], max_floats * 4));
    CK(hipMemset(buf[d], d, max_floats * 4));
    CK(hipStreamCreateWithFlags(&st[d], hipStreamNonBlocking));
    for (int p = 0; p < ndev; p++) {
      if (p == d) continue;
      int can = 0;
      CK(hipDeviceCanAccessPeer(&can, d, p));
      if (can) {
        hipError_t e = hipDeviceEnablePeerAccess(p, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) CK(e);
        (void)hipGetLastError();
      } else {
        printf("device %d cannot access device %d directly\n", d, p);
      }
    }
  }
  auto sync_all = [&] {
    for (int d = 0; d < ndev; d++) {
      CK(hipSetDevice(d));
      CK(hipDeviceSynchronize());
    }
  };
  printf("# %d devices.  Totals are the whole gathered vector; a slice is total / ndev\n", ndev);
  // ---- 1a: one peer copy at a time, device 0 -> d
  for (size_t mb : totals_mb) {
    const size_t slice = (mb << 20) / ndev;
    for (int d = 1; d < ndev; d++) {
      sync_all();
      double best = 1e30;
      for (int rep = 0; rep < 5; rep++) {
        const double t0 = now_ms();
        CK(hipMemcpyPeerAsync(buf[d], d, buf[0], 0, slice, st[0]));
        CK(hipStreamSynchronize(st[0]));
        const double t = now_ms() - t0;
        best = t < best ? t : best;
      }
      printf("peer copy 0 -> %d  slice %7.2f MB: %7.3f ms = %6.1f GB/s\n", d, slice / 1048576.0, best, slice / best / 1e6);
    }
  }
  // ---- 1b: all pairs at once: every device pulls the slices of all others (what the p2p exchange of gdn_pr_multi does)
  for (size_t mb : totals_mb) {
    const size_t slice = (mb << 20) / ndev;
    double best = 1e30;
    for (int rep = 0; rep < 5; rep++) {
      sync_all();
      const double t0 = now_ms();
      for (int d = 0; d < ndev; d++)
        for (int k = 1; k < ndev; k++) {
          const int p = (d + k) % ndev;  // staggered partners: no two devices start on the same source
          CK(hipMemcpyPeerAsync(reinterpret_cast<char *>(buf[d]) + (size_t)p * slice, d,
                                reinterpret_cast<char *>(buf[p]) + (size_t)p * slice, p, slice, st[d]));
        }
      for (int d = 0; d < ndev; d++) CK(hipStreamSynchronize(st[d]));
      const double t = now_ms() - t0;
      best = t < best ? t : best;
    }
    const double recv = (double)slice * (ndev - 1);
    printf("all pairs, total %4zu MB: %7.3f ms; every device receives %6.1f MB = %6.1f GB/s in (%5.1f per link if %d links run side by side)\n",
           mb, best, recv / 1048576.0, recv / best / 1e6, recv / best / 1e6 / (ndev - 1), ndev - 1);
  }
  // ---- 2: RCCL in-place all-gather, one thread per device
  std::vector<ncclComm_t> comm(ndev);
  std::vector<int> devs(ndev);
  for (int d = 0; d < ndev; d++) devs[d] = d;
  NK(ncclCommInitAll(comm.data(), ndev, devs.data()));
  for (size_t mb : totals_mb) {
    const size_t slice_f = (mb << 20) / 4 / ndev;
    double best = 1e30;
    for (int rep = 0; rep < 6; rep++) {
      sync_all();
      const double t0 = now_ms();
      NK(ncclGroupStart());
      for (int d = 0; d < ndev; d++) {
        CK(hipSetDevice(d));
        NK(ncclAllGather(buf[d] + (size_t)d * slice_f, buf[d], slice_f, ncclFloat, comm[d], st[d]));
      }
      NK(ncclGroupEnd());
      for (int d = 0; d < ndev; d++) {
        CK(hipSetDevice(d));
        CK(hipStreamSynchronize(st[d]));
      }
      const double t = now_ms() - t0;
      if (rep) best = t < best ? t : best;  // the first call builds RCCL's channels
    }
    const double recv = (double)slice_f * 4 * (ndev - 1);
    printf("ncclAllGather, total %4zu MB: %7.3f ms; every device receives %6.1f MB = %6.1f GB/s in\n", mb, best, recv / 1048576.0,
           recv / best / 1e6);
  }
  for (int d = 0; d < ndev; d++) ncclCommDestroy(comm[d]);
  printf("# direct mesh: the all-pairs and all-gather rates should be ~ (ndev - 1) x the single-link rate of 1a; a ring shows ~ 1 x\n");
  return 0;
}
