// gdn_common.hpp -- shared host/device plumbing of libgardenia_hip.so (gfx950 only).
//
// Device primitives here are the hand-written wave64 replacements for what the reference
// takes from CUB/Thrust: cub::BlockScan<int,256>::ExclusiveSum
// (include/worklistc.h:73, src/bfs/linear_lb.cu:152), cub::BlockReduce (src/pr/base.cu:48,
// src/tc/gpu_base.cu:22) and the Worklist2 push (include/worklistc.h:66-113).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/gardenia_hip.h"

#define GDN_WAVE 64
#define GDN_BLOCK 256                 // threads per workgroup for every kernel (4 waves)
#define GDN_WAVES_PER_BLOCK (GDN_BLOCK / GDN_WAVE)
#define GDN_MYINFINITY 1000000000     // include/common.h:66
#define GDN_DIST_INF 2147483647       // src/sssp/sssp.h:46 kDistInf as int32

typedef uint64_t eoff_t;  // edge offsets: never held in 32 bits (RMAT-27 has 2^31 edges)
typedef int32_t vid_t;

// ------------------------------------------------------------------------------------------
// host-side error plumbing: no exit(), thread-local message
// ------------------------------------------------------------------------------------------
void gdn_set_error(const char *fmt, ...);

#define GDN_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e__ = (call);                                                                   \
    if (e__ != hipSuccess) {                                                                   \
      gdn_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__));     \
      return (e__ == hipErrorOutOfMemory) ? GDN_ERR_OOM                                        \
             : (e__ == hipErrorNoDevice || e__ == hipErrorInvalidDevice) ? GDN_ERR_NO_DEVICE   \
                                                                         : GDN_ERR_HIP;        \
    }                                                                                          \
  } while (0)

#define GDN_TRY(call)               \
  do {                              \
    int s__ = (call);               \
    if (s__ != GDN_OK) return s__;  \
  } while (0)

#define GDN_REQUIRE(cond, msg)                                       \
  do {                                                               \
    if (!(cond)) {                                                   \
      gdn_set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, msg); \
      return GDN_ERR_INVALID;                                        \
    }                                                                \
  } while (0)

int gdn_require_device();
// value of a library option: the environment variable `name` if set, else what gdn_option_set stored, else nullptr
// Three classes of run-time options:
//   gdn_option       PUBLIC options -- include/gardenia_hip.h lists every one of them next to gdn_option_set
//   gdn_test_option  TEST HOOKS (thresholds that force a big-graph code path onto a small graph, layout variants the suite
//                    compares bit for bit): honoured only while GDN_TEST_HOOKS=1 is set (tests/conftest.py sets it); a
//                    production process does not read them
//   gdn_xoption      A/B knobs of closed experiments and measurement sessions: compiled in only with -DGDN_EXPERIMENTS
//                    (make EXPERIMENTS=1, tools/build_variant.sh); nullptr -- and their code paths dead -- in the shipped build
const char *gdn_option(const char *name);
const char *gdn_test_option(const char *name);
#ifdef GDN_EXPERIMENTS
#define gdn_xoption(name) gdn_option(name)
#else
#define gdn_xoption(name) (static_cast<const char *>(nullptr))
#endif

// Byte offset for the next large allocation (gdn_graph.hip).  0 unless the option GDN_ALLOC_STAGGER names a granule
// (a multiple of 256 bytes): then the k-th buffer of >= 1 MiB starts (2k + 1) mod 127 granules behind its hipMalloc
// base -- an A/B knob for the placement spread of DESIGN.md 4.1 (streams that start at power-of-two-aligned bases
// walk the memory channels in step).
size_t gdn_alloc_stagger_next(size_t bytes);
hipError_t gdn_plain_malloc(void **p, size_t bytes);  // hipMalloc / hipFree, fenced under GDN_ALLOC_FENCE=1 (gdn_graph.hip)
hipError_t gdn_plain_free(void *p);

// Short-lived device memory of a build (sort keys, the layout builder's arenas): a process-level cache of hipMalloc blocks
// (gdn_graph.hip), NOT hipMalloc / hipFree per use.  Measured (profiles/r04_malloc_probe.txt, r04 sessions 3-5): a hipMalloc
// of 16 GB takes 0.8 - 4.8 s whenever memory of that size was freed shortly before (the driver hands freed pages out again
// only after wiping them), and hipFree costs 0.16 ms of synchronisation whatever the size.  A freed block may still be in
// use by work queued on the null stream; its next user queues behind that work.
void *gdn_reserve_take(size_t bytes);  // the block gdn_dev_reserve set aside, if it holds `bytes` (then the caller's; hipFree)
int gdn_scratch_malloc(void **p, size_t bytes, int site = 8);  // gdn_graph.hip (site: bit of GDN_SCRATCH_POISON_SITES)
void gdn_scratch_free(void *p);
void gdn_scratch_trim();  // the cache goes back to the driver

// RAII device buffer (solver-private scratch)
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  void *base = nullptr;  // what hipMalloc returned (p may sit a staggered offset behind it)
  bool pooled = false;   // base came from gdn_scratch_malloc
  DevBuf() {}
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (base) {
      if (pooled) gdn_scratch_free(base);
      else (void)hipFree(base);
    }
    base = nullptr;
    p = nullptr;
    n = 0;
    pooled = false;
  }
  // a temporary of a build (see gdn_scratch_malloc); never one of a plan's long-lived arrays
  int alloc_scratch(size_t count) {
    release();
    n = count;
    if (count == 0) count = 1;
    GDN_TRY(gdn_scratch_malloc(&base, count * sizeof(T)));
    p = static_cast<T *>(base);
    pooled = true;
    return GDN_OK;
  }
  int alloc(size_t count) {
    release();
    n = count;
    if (count == 0) count = 1;
    const size_t off = gdn_alloc_stagger_next(count * sizeof(T));
    hipError_t e = hipMalloc(&base, count * sizeof(T) + off);
    if (e != hipSuccess) {  // the scratch cache may hold what is missing
      (void)hipGetLastError();
      gdn_scratch_trim();
      e = hipMalloc(&base, count * sizeof(T) + off);
    }
    if (e != hipSuccess) {
      base = nullptr;
      p = nullptr;
      gdn_set_error("hipMalloc(%zu bytes) -> %s", count * sizeof(T) + off, hipGetErrorString(e));
      return GDN_ERR_OOM;
    }
    p = reinterpret_cast<T *>(static_cast<char *>(base) + off);
    return GDN_OK;
  }
  // an allocation the caller can do without (a placement candidate): no trim of the scratch cache, no error text, the HIP
  // last-error cleared on failure, and refused while it would take more than half of the device's free memory
  bool alloc_optional(size_t count) {
    release();
    if (count == 0) count = 1;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || count * sizeof(T) > free_b / 2) {
      (void)hipGetLastError();
      return false;
    }
    if (hipMalloc(&base, count * sizeof(T)) != hipSuccess) {
      (void)hipGetLastError();
      base = nullptr;
      return false;
    }
    n = count;
    p = static_cast<T *>(base);
    return true;
  }
  // the same contents in a FRESH allocation (made while the old one is still held, so it is other memory); `keep` gets the
  // old allocation instead of hipFree when the caller may want to go back (gdn_pr_plan_place: where hipMalloc puts a
  // streamed array moves a PageRank iteration by up to 8 %)
  int move(DevBuf<T> *keep = nullptr) {
    if (!p || n == 0) return GDN_OK;
    void *nb = nullptr;
    hipError_t e = hipMalloc(&nb, n * sizeof(T));
    if (e != hipSuccess) {
      (void)hipGetLastError();
      gdn_set_error("hipMalloc(%zu bytes) -> %s", n * sizeof(T), hipGetErrorString(e));
      return GDN_ERR_OOM;
    }
    e = hipMemcpy(nb, p, n * sizeof(T), hipMemcpyDeviceToDevice);
    if (e != hipSuccess) {
      (void)hipFree(nb);
      gdn_set_error("move copy -> %s", hipGetErrorString(e));
      return GDN_ERR_HIP;
    }
    if (keep) {
      keep->release();
      keep->base = base;
      keep->p = p;
      keep->n = n;
      keep->pooled = pooled;
    } else if (pooled) {
      gdn_scratch_free(base);
    } else {
      (void)hipFree(base);
    }
    base = nb;
    p = static_cast<T *>(nb);
    pooled = false;
    return GDN_OK;
  }
  void swap(DevBuf<T> &o) {
    T *tp = p; p = o.p; o.p = tp;
    size_t tn = n; n = o.n; o.n = tn;
    void *tb = base; base = o.base; o.base = tb;
    bool tq = pooled; pooled = o.pooled; o.pooled = tq;
  }
#ifdef GDN_EXPERIMENTS  // placement A/B knobs of measurement builds only (make EXPERIMENTS=1, tools/build_variant.sh)
  // move the contents into an allocation made with hipExtMallocWithFlags(flags) (A/B knob: hipDeviceMallocUncached keeps a
  // read-once stream out of the XCD L2s)
  int rehome(unsigned flags) {
    if (!p || n == 0) return GDN_OK;
    void *nb = nullptr;
    hipError_t e = hipExtMallocWithFlags(&nb, n * sizeof(T), flags);
    if (e != hipSuccess) {
      gdn_set_error("hipExtMallocWithFlags(%zu bytes, %u) -> %s", n * sizeof(T), flags, hipGetErrorString(e));
      return GDN_ERR_OOM;
    }
    e = hipMemcpy(nb, p, n * sizeof(T), hipMemcpyDeviceToDevice);
    if (e != hipSuccess) {
      (void)hipFree(nb);
      gdn_set_error("rehome copy -> %s", hipGetErrorString(e));
      return GDN_ERR_HIP;
    }
    (void)hipFree(base);
    base = nb;
    p = static_cast<T *>(nb);
    return GDN_OK;
  }
  // move the contents into ONE virtual range backed by physical chunks of `chunk` bytes mapped in a shuffled order
  // (hipMemCreate / hipMemMap; A/B knob of the placement spread).  The range is never unmapped: measurement
  // processes only (GDN_EXPERIMENTS builds).
  int rehome_shuffled(size_t chunk, unsigned seed) {
    if (!p || n == 0) return GDN_OK;
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) gran = 2u << 20;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t bytes = n * sizeof(T), nch = (bytes + chunk - 1) / chunk, total = nch * chunk;
    void *va = nullptr;
    if (hipMemAddressReserve(&va, total, chunk, nullptr, 0) != hipSuccess) {
      gdn_set_error("hipMemAddressReserve(%zu) failed", total);
      return GDN_ERR_OOM;
    }
    std::vector<size_t> perm(nch);
    for (size_t i = 0; i < nch; i++) perm[i] = i;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (seed + 1);
    for (size_t i = nch; i > 1; i--) {  // Fisher-Yates with a xorshift
      x ^= x << 13;
      x ^= x >> 7;
      x ^= x << 17;
      const size_t j = (size_t)(x % i);
      const size_t t = perm[i - 1];
      perm[i - 1] = perm[j];
      perm[j] = t;
    }
    for (size_t i = 0; i < nch; i++) {  // physical chunks are created in order and land at shuffled virtual offsets
      hipMemGenericAllocationHandle_t h;
      if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess ||
          hipMemMap(static_cast<char *>(va) + perm[i] * chunk, chunk, 0, h, 0) != hipSuccess) {
        gdn_set_error("hipMemCreate / hipMemMap of chunk %zu failed", i);
        return GDN_ERR_OOM;
      }
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) {
      gdn_set_error("hipMemSetAccess failed");
      return GDN_ERR_HIP;
    }
    if (hipMemcpy(va, p, bytes, hipMemcpyDeviceToDevice) != hipSuccess) {
      gdn_set_error("rehome_shuffled copy failed");
      return GDN_ERR_HIP;
    }
    (void)hipFree(base);
    base = nullptr;  // (leaked on purpose: see above)
    p = static_cast<T *>(va);
    return GDN_OK;
  }
#endif
  void take(DevBuf &o) {  // this buffer takes o's memory over
    release();
    p = o.p;
    n = o.n;
    base = o.base;
    pooled = o.pooled;
    o.p = nullptr;
    o.base = nullptr;
    o.n = 0;
    o.pooled = false;
  }
};

// ---- GdnMailbox: a small device struct read back by the host WITHOUT a stream synchronisation.  A level-synchronous
// solver reads a few counters per level; hipMemcpyAsync + hipStreamSynchronize costs 16-18 us of idle GPU per level
// (rocprofv3 --kernel-trace of a search, profiles/r03_bfs_bottom_up.txt).  Here a one-wave kernel, queued behind the level's
// kernels, copies the struct into pinned host memory and then publishes a sequence number with system scope; the host spins
// on the number.  GDN_MAILBOX=0 (or no pinned memory) falls back to the copy; a spin of more than two seconds falls back to
// the synchronisation too, which then reports what went wrong.
// zero != 0: the struct is zeroed behind the copy -- counters that the next level adds to again need no memset of their own (a
// dispatch per level less)
static __global__ void gdn_mailbox_kernel(unsigned *__restrict__ src, unsigned nwords, unsigned *dst, unsigned seq, int zero) {
  for (unsigned i = threadIdx.x; i < nwords; i += 64) {
    __hip_atomic_store(dst + 1 + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (zero) src[i] = 0u;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(dst, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct GdnMailbox {
  static constexpr size_t kPayload = 1024;  // bytes
  unsigned *host = nullptr;  // pinned: [0] sequence number, [1 ..] payload
  unsigned *dev = nullptr;   // the same memory as the device sees it
  unsigned seq = 0;
  bool spin = true;
  ~GdnMailbox() {
    if (host) (void)hipHostFree(host);
  }
  void init() {
    if (host) return;
    const char *e = gdn_xoption("GDN_MAILBOX");
    spin = !(e && e[0] == '0');
    void *h = nullptr;
    if (hipHostMalloc(&h, kPayload + 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipHostFree(h);
      return;
    }
    host = static_cast<unsigned *>(h);
    dev = static_cast<unsigned *>(d);
    host[0] = 0;
  }
  template <typename T>
  int read(T *d_src, T &out, hipStream_t s = 0, bool zero = false) {
    static_assert(sizeof(T) <= kPayload && sizeof(T) % 4 == 0, "mailbox payload");
    if (!host) {
      GDN_HIP(hipMemcpy(&out, d_src, sizeof(T), hipMemcpyDeviceToHost));
      if (zero) GDN_HIP(hipMemsetAsync(d_src, 0, sizeof(T), s));
      return GDN_OK;
    }
    if (!spin) {
      GDN_HIP(hipMemcpyAsync(host + 1, d_src, sizeof(T), hipMemcpyDeviceToHost, s));
      GDN_HIP(hipStreamSynchronize(s));
      memcpy(&out, host + 1, sizeof(T));
      if (zero) GDN_HIP(hipMemsetAsync(d_src, 0, sizeof(T), s));
      return GDN_OK;
    }
    ++seq;
    if (seq == 0) ++seq;
    hipLaunchKernelGGL(gdn_mailbox_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned *>(d_src),
                       (unsigned)(sizeof(T) / 4), dev, seq, zero ? 1 : 0);
    GDN_HIP(hipGetLastError());
    const volatile unsigned *flag = host;
    unsigned long long spins = 0;
    auto t0 = std::chrono::steady_clock::now();
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
      if ((++spins & 0xFFFFu) == 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
        GDN_HIP(hipStreamSynchronize(s));  // reports a failed kernel; else the number is there now
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
          gdn_set_error("mailbox: the sequence number never arrived");
          return GDN_ERR_HIP;
        }
        break;
      }
    }
    memcpy(&out, const_cast<const unsigned *>(host + 1), sizeof(T));
    return GDN_OK;
  }
};

struct HostTimer {
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t s;
  explicit HostTimer(hipStream_t st = 0) : s(st) {
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
  }
  ~HostTimer() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
  void start() { (void)hipEventRecord(a, s); }
  double stop_ms() {
    (void)hipEventRecord(b, s);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return (double)ms;
  }
};

struct gdn_graph {
  int32_t m = 0;
  uint64_t nnz = 0;
  eoff_t *rowptr = nullptr;  // device, m+1
  vid_t *colidx = nullptr;   // device, nnz
  bool owned = false;
};

static inline unsigned gdn_nblocks(uint64_t n, unsigned per_block = GDN_BLOCK) {
  uint64_t b = (n + per_block - 1) / per_block;
  return (unsigned)(b == 0 ? 1 : b);
}

// device-wide helpers implemented in gdn_graph.hip
int gdn_exclusive_scan_u32_to_u64(const uint32_t *d_in, eoff_t *d_out, size_t n, hipStream_t s);
int gdn_fill_i32(int32_t *d, int32_t v, size_t n, hipStream_t s);
void gdn_pr_trace_set(const double *diff, int32_t n);  // gdn_pr.hip: the trace gdn_pr_last_trace returns
int gdn_graph_pad_cols(gdn_graph *s, int32_t world, const int32_t *bounds, int32_t chunk);

#ifdef __HIPCC__
// ------------------------------------------------------------------------------------------
// wave64 primitives
// ------------------------------------------------------------------------------------------
// IEEE single-precision operations the compiler must NOT contract into an fma.  hipcc compiles with
// -ffp-contract=fast-honor-pragmas and the __fmul_rn / __fadd_rn of <__clang_hip_math.h> are plain operators, so
// __fadd_rn(score, __fmul_rn(d, sum)) became ONE v_fmac_f32 -- a single rounding where the reference's CPU solvers
// (g++ -O3 for baseline x86-64: no fma) round twice.  Wherever a result is compared with the reference operation by
// operation, multiply with gdn_fmul: a product without the `contract` flag cannot be fused into the add that uses it.
__device__ __forceinline__ float gdn_fmul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float gdn_fadd(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float gdn_fsub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

__device__ __forceinline__ unsigned gdn_lane() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Ops with `static constexpr bool kSkippable = true` and a member `const unsigned *skip` (PageRank): a launch whose
// *skip is non-zero returns at once -- the host queues a BATCH of iterations without reading the L1 change back after
// each, a one-thread kernel behind every iteration sets the flag at convergence (gdn_pr.hip), and the iterations queued
// behind it leave the state alone.  Other ops: no code at all.
template <class T, class = void>
struct GdnSkippable {
  static constexpr bool value = false;
};
template <class T>
struct GdnSkippable<T, decltype((void)T::kSkippable)> {
  static constexpr bool value = T::kSkippable;
};
template <class Op>
__device__ __forceinline__ bool gdn_skip_launch(const Op &op) {
  if constexpr (GdnSkippable<Op>::value) return op.skip != nullptr && *op.skip != 0u;
  return false;
}

// Level boundary inside ONE workgroup (the fused light-level kernels): all its waves go through the same vector L1 and the
// same L2, so waiting for the outstanding accesses and a barrier orders them.  __threadfence() is the device-scope form --
// buffer_wbl2 sc1 + buffer_inv sc1 on gfx950: a write-back of the L2 and an invalidate of the L1 per level, microseconds
// each, that only a reader on another XCD needs.  What ATOMICS wrote (they execute in the L2) is still read past the L1
// with device-scope atomic loads.
__device__ __forceinline__ void gdn_wg_level_sync() {
#ifdef GDN_ABL_LEVEL_FENCE  // A/B builds only (tools/build_variant.sh): the device-scope fence this replaced
  __threadfence();
#else
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
  __syncthreads();
}

// Grid barrier of a cooperative launch (all workgroups co-resident), for kernels that touch their shared MUTABLE data
// with device-scope accesses only (atomics, `sc1` loads and stores -- the queues, counters, distances and bitmaps of the
// persistent traversal kernels): every wave drains its own accesses, the workgroup meets, one lane takes a TICKET on
// bar[0] (never reset during a launch), the workgroup whose ticket completes a generation publishes that generation's
// number on bar[32] (a line of its own), the others poll it.  A workgroup derives the generation it waits for from its
// OWN ticket, so nothing has to be read before arriving: the first form sampled the generation word with a relaxed
// load and then arrived on another line, and had the arrival been performed first the last arriver could have bumped
// the generation in between -- this workgroup would have polled for ever.  No cache write-back /
// invalidate is needed (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores drained before the signal + sc1
// loads behind it); three __threadfence() per barrier, as first written, cost a third of a light level (4096 x 4096
// lattice BFS: 15.6 -> 10.0 us per level).  Arrivals per XCD with a second level on top (eight counters in parallel,
// XCD read from HW_REG_XCC_ID) measured the same 9.9 us: the level is bound by its chain of dependent accesses, not
// by the 256 arrivals -- the flat form stays.  Tickets wrap after 2^32 arrivals per launch (16 M barriers of a
// 256-workgroup grid); the comparison below is wrap-safe anyway.
#define GDN_GBAR_WORDS 64  // zeroed by the host before every launch
__device__ __forceinline__ void gdn_grid_barrier(unsigned *bar, unsigned nblocks) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(bar, 1u);
    const unsigned gen = t / nblocks + 1u;  // the generation this arrival belongs to (1, 2, ...)
    if (t % nblocks == nblocks - 1u) {
      __hip_atomic_store(bar + 32, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while ((int)(__hip_atomic_load(bar + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - gen) < 0)
        __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

__device__ __forceinline__ unsigned long long gdn_lanemask_lt() {
  return (1ull << gdn_lane()) - 1ull;
}

template <typename T>
__device__ __forceinline__ T gdn_wave_sum(T v) {  // every lane gets the total
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T>
__device__ __forceinline__ T gdn_wave_incl_scan(T v) {
  const unsigned lane = gdn_lane();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T p = __shfl_up(v, o, 64);
    if (lane >= (unsigned)o) v += p;
  }
  return v;
}

// Block (256 threads) sum; result valid in every thread.  `scratch` = GDN_WAVES_PER_BLOCK Ts.
template <typename T>
__device__ __forceinline__ T gdn_block_sum(T v, T *scratch) {
  v = gdn_wave_sum(v);
  const unsigned w = threadIdx.x >> 6;
  __syncthreads();
  if (gdn_lane() == 0) scratch[w] = v;
  __syncthreads();
  T t = scratch[0];
#pragma unroll
  for (int i = 1; i < GDN_WAVES_PER_BLOCK; i++) t += scratch[i];
  return t;
}

// Block exclusive scan; *total gets the block sum.  scratch = GDN_WAVES_PER_BLOCK Ts.
// (replaces cub::BlockScan<int,256>::ExclusiveSum, include/worklistc.h:73)
template <typename T>
__device__ __forceinline__ T gdn_block_excl_scan(T v, T *scratch, T *total) {
  T incl = gdn_wave_incl_scan(v);
  const unsigned w = threadIdx.x >> 6;
  __syncthreads();
  if (gdn_lane() == 63) scratch[w] = incl;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < GDN_WAVES_PER_BLOCK; i++) {
    T s = scratch[i];
    if ((unsigned)i < w) base += s;
    tot += s;
  }
  *total = tot;
  return base + incl - v;
}

// Wavefront-aggregated worklist push: ONE atomicAdd per wave, lanes take consecutive slots
// by ballot + popcount prefix (replaces Worklist::push, include/worklistc.h:44-50 -- one
// atomicAdd per item -- and Worklist2::push_1item's CUB block scan, :66-89).  Must be called
// by every active lane of the wave with the same queue.  Overflow is reported through
// *overflow instead of being dropped (worklistc.h:46-47 drops silently).
__device__ __forceinline__ void gdn_wl_push(vid_t *queue, unsigned *count, unsigned capacity,
                                            bool pred, vid_t item, unsigned *overflow) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0ull) return;
  const unsigned lane = gdn_lane();
  const int leader = __ffsll((long long)mask) - 1;
  unsigned base = 0;
  if ((int)lane == leader) base = atomicAdd(count, (unsigned)__popcll(mask));
  base = __shfl(base, leader, 64);
  if (pred) {
    const unsigned pos = base + (unsigned)__popcll(mask & gdn_lanemask_lt());
    if (pos < capacity) queue[pos] = item;
    else *overflow = 1u;
  }
}

// Staged form of gdn_wl_push: items are collected in a per-wave LDS strip and flushed with ONE atomicAdd per
// GDN_WL_STAGE - 64 items or so.  A single hot counter takes ~12 ns per atomic on this chip whatever issues it
// (a top-down BFS level that pushed from 87 K wave steps spent 1 ms of its 1.5 ms there); staging divides the
// count by ~4-16.  `n` is the wave-uniform fill of the strip; call gdn_wl_flush at the end of the kernel.
#ifndef GDN_WL_STAGE
#define GDN_WL_STAGE 256
#endif
struct GdnWlStage {
  vid_t *strip;  // GDN_WL_STAGE entries of LDS owned by this wave
  unsigned n;
};

__device__ __forceinline__ void gdn_wl_flush(GdnWlStage &st, vid_t *queue, unsigned *count, unsigned capacity,
                                             unsigned *overflow) {
  if (st.n == 0) return;
  const unsigned lane = gdn_lane();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  unsigned base = 0;
  if (lane == 0) base = atomicAdd(count, st.n);
  base = __shfl(base, 0, 64);
  for (unsigned i = lane; i < st.n; i += 64) {
    if (base + i < capacity) queue[base + i] = st.strip[i];
    else *overflow = 1u;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  st.n = 0;
}

// The LAST flush of a kernel, by the whole workgroup (every thread calls it at the end, after its pushes): the strips of the
// workgroup's waves go out behind ONE reservation.  A persistent grid of 2048 workgroups otherwise ends with 8192 partial
// flushes on one counter (~12-25 ns each, one after the other: 0.1-0.2 ms whatever the kernel did before).
// s_tmp: GDN_WAVES_PER_BLOCK + 1 unsigned of LDS.
__device__ __forceinline__ void gdn_wl_flush_block(GdnWlStage &st, vid_t *queue, unsigned *count, unsigned capacity,
                                                   unsigned *overflow, unsigned *s_tmp) {
  const unsigned lane = gdn_lane(), w = threadIdx.x >> 6;
  __syncthreads();  // (s_tmp may have held something else; the strips are complete)
  if (lane == 0) s_tmp[w] = st.n;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned total = 0;
    for (unsigned i = 0; i < GDN_WAVES_PER_BLOCK; i++) total += s_tmp[i];
    s_tmp[GDN_WAVES_PER_BLOCK] = total ? atomicAdd(count, total) : 0u;
  }
  __syncthreads();
  unsigned base = s_tmp[GDN_WAVES_PER_BLOCK];
  for (unsigned i = 0; i < w; i++) base += s_tmp[i];
  for (unsigned i = lane; i < st.n; i += 64) {
    if (base + i < capacity) queue[base + i] = st.strip[i];
    else *overflow = 1u;
  }
  st.n = 0;
  __syncthreads();
}

// sum of a per-thread value over the workgroup, added to a global counter by ONE thread (s_tmp: GDN_WAVES_PER_BLOCK u64)
__device__ __forceinline__ void gdn_block_add_u64(unsigned long long v, unsigned long long *counter, unsigned long long *s_tmp) {
  v = gdn_wave_sum(v);
  __syncthreads();
  if (gdn_lane() == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (unsigned i = 0; i < GDN_WAVES_PER_BLOCK; i++) t += s_tmp[i];
    if (t) atomicAdd(counter, t);
  }
}

// maximum of a per-thread unsigned over the workgroup, atomicMax-ed into a global word by ONE thread
__device__ __forceinline__ void gdn_block_max_u32(unsigned v, unsigned *out) {
  __shared__ unsigned s_mx[GDN_WAVES_PER_BLOCK];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t = (unsigned)__shfl_xor((int)v, o, 64);
    v = t > v ? t : v;
  }
  if (gdn_lane() == 0) s_mx[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (unsigned i = 1; i < GDN_WAVES_PER_BLOCK; i++) v = s_mx[i] > v ? s_mx[i] : v;
    if (v) atomicMax(out, v);
  }
}

// must be called by every active lane of the wave (convergent), like gdn_wl_push
__device__ __forceinline__ void gdn_wl_push_staged(GdnWlStage &st, vid_t *queue, unsigned *count, unsigned capacity,
                                                   bool pred, vid_t item, unsigned *overflow) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0ull) return;
  const unsigned c = (unsigned)__popcll(mask);
  if (st.n + c > GDN_WL_STAGE) gdn_wl_flush(st, queue, count, capacity, overflow);
  if (pred) st.strip[st.n + (unsigned)__popcll(mask & gdn_lanemask_lt())] = item;
  st.n += c;
}
#endif  // __HIPCC__
