// gdn_graph.hip -- resident CSR graphs, error plumbing and the device-wide scan.
//
// Replaces the per-call cudaMalloc/cudaMemcpy of the whole CSR at the top of every reference
// solver (src/bfs/linear_base.cu:42-51, src/pr/base.cu:83-99) with an explicit handle, so
// iterations can be timed with the graph resident (SURVEY 8b "resident-graph variant").
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "gdn_common.hpp"

static thread_local char g_err[512] = "";

// ---- options: every tuning / test knob of the library ("GDN_...") is an OPTION that a caller sets through the API
// (gdn_option_set) and the environment variable of the same name OVERRIDES (so measurements and tests can flip a knob
// without touching the caller).  Values are strings, parsed where they are used, exactly like the environment's.
#include <algorithm>
#include <atomic>
#include <iterator>
#include <map>
#include <mutex>
static std::mutex g_opt_mu;
static std::map<std::string, std::string> &gdn_opt_map() {
  static std::map<std::string, std::string> m;
  return m;
}
const char *gdn_option(const char *name) {
  if (const char *e = getenv(name)) return e;
  std::lock_guard<std::mutex> lk(g_opt_mu);
  auto &m = gdn_opt_map();
  auto it = m.find(name);
  return it == m.end() ? nullptr : it->second.c_str();  // (the string lives until the option is set again)
}

const char *gdn_test_option(const char *name) {
  const char *h = gdn_option("GDN_TEST_HOOKS");
  return (h && h[0] == '1') ? gdn_option(name) : nullptr;
}

// GDN_ALLOC_FENCE=1 (debugging, read once): every DevBuf / scratch allocation is its own block of whole 2 MB pages with the
// buffer at its END (16-byte aligned), and the scratch cache is off -- a kernel that reads or writes past the end of a
// buffer then leaves the mapping and faults on the spot instead of touching a neighbour (the GPU build has no address
// sanitizer on this pool).  tests/aids/fuzz_parity.py and the parity suite run unchanged under it, only slower.
bool gdn_alloc_fence() {
  static const bool on = [] {
    const char *e = gdn_option("GDN_ALLOC_FENCE");
    return e && e[0] == '1';
  }();
  return on;
}
static size_t gdn_fence_offset(size_t bytes) {
  const size_t page = (size_t)2 << 20, b16 = (bytes + 15) & ~(size_t)15;
  return ((b16 + page - 1) & ~(page - 1)) - b16;
}

size_t gdn_alloc_stagger_next(size_t bytes) {
  static std::atomic<unsigned> k{0};
  if (gdn_alloc_fence()) return gdn_fence_offset(bytes);
  if (bytes < (1u << 20)) return 0;
  const char *e = gdn_xoption("GDN_ALLOC_STAGGER");
  if (!e) return 0;
  const size_t g = (size_t)strtoull(e, nullptr, 10) & ~(size_t)255;
  if (g == 0) return 0;
  return (size_t)((2u * k.fetch_add(1u) + 1u) % 127u) * g;
}

void gdn_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- scratch cache (gdn_common.hpp): freed scratch blocks stay with the process, keyed by device and size, and are handed
// out again to requests of about their size.  Plain hipMalloc pointers: the stream-ordered pool (hipMallocAsync) was
// measured first and dropped -- on this runtime (HIP 7.0.51831) a hipMemsetAsync on pool memory is NOT ordered in front of
// the kernel queued behind it (tools/memset_probe.hip, profiles/r04_memset_probe.txt: 16 896 bytes zeroed AFTER the kernel
// had set its bits, deterministically; never on hipMalloc memory), and two fuzz sweeps failed on it.
// GDN_SCRATCH_POOL=0 (read once): plain hipMalloc / hipFree instead (A/B knob)
static bool gdn_scratch_pooled() {
  static const bool on = [] {
    const char *e = gdn_xoption("GDN_SCRATCH_POOL");
    return !(e && e[0] == '0');
  }();
  return on;
}
namespace {
struct ScratchCache {
  static constexpr int kDevs = 16;
  std::mutex mu;
  std::multimap<size_t, void *> free_[kDevs];  // size -> block
  std::map<void *, std::pair<int, size_t>> live;  // block -> (device, size)
  size_t cached[kDevs] = {};
  size_t keep = 32ull << 30;  // cached bytes per device beyond which the largest blocks go back (GDN_SCRATCH_KEEP_GB)
  ScratchCache() {
    if (const char *e = gdn_option("GDN_SCRATCH_KEEP_GB")) keep = (size_t)strtoull(e, nullptr, 10) << 30;
  }
};
ScratchCache &scratch_cache() {
  static ScratchCache *c = new ScratchCache();  // never destroyed: the HIP runtime may be gone before static destructors run
  return *c;
}
// size classes: eighths of a power of two (at most 12.5 % over the request), never below 4 KB
size_t scratch_round(size_t bytes) {
  if (bytes < 4096) return 4096;
  size_t p2 = 4096;
  while ((p2 << 1) <= bytes) p2 <<= 1;
  const size_t g = p2 >> 3;
  return (bytes + g - 1) / g * g;
}
}  // namespace

// everything the cache holds goes back to the driver (an allocation failed, or a caller wants the memory)
void gdn_scratch_trim() {
  ScratchCache &c = scratch_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (int d = 0; d < ScratchCache::kDevs; d++) {
    if (c.free_[d].empty()) continue;
    (void)hipSetDevice(d);
    for (auto &kv : c.free_[d]) (void)hipFree(kv.second);
    c.free_[d].clear();
    c.cached[d] = 0;
  }
  (void)hipSetDevice(cur);
}

// GDN_SCRATCH_POISON=1 (read once): every scratch block starts as 0xCD bytes -- finds code that counts on fresh device
// memory being zero (it is, from hipMalloc; a block of the pool holds whatever its last user left)
static bool gdn_scratch_poison() {
  static const bool on = [] {
    const char *e = gdn_test_option("GDN_SCRATCH_POISON");
    return e && e[0] == '1';
  }();
  return on;
}
static int gdn_scratch_malloc_raw(void **p, size_t bytes);
int gdn_scratch_malloc(void **p, size_t bytes, int site) {
  GDN_TRY(gdn_scratch_malloc_raw(p, bytes));
  if (gdn_scratch_poison()) {
    static const int sites = [] {
      const char *e = gdn_xoption("GDN_SCRATCH_POISON_SITES");  // bit mask of the sites to poison (debugging)
      return e ? atoi(e) : 0xFFFF;
    }();
    if (sites & site) {
      // GDN_SCRATCH_POISON_PART (debugging): "head:N" only the first N bytes, "tail:N" only the last N, "mid:A:B" bytes [A, B)
      size_t lo = 0, hi = bytes ? bytes : 1;
      if (const char *e = gdn_xoption("GDN_SCRATCH_POISON_PART")) {
        if (!strncmp(e, "head:", 5)) hi = std::min(hi, (size_t)strtoull(e + 5, nullptr, 10));
        else if (!strncmp(e, "tail:", 5)) lo = hi - std::min(hi, (size_t)strtoull(e + 5, nullptr, 10));
        else if (!strncmp(e, "mid:", 4)) {
          char *q = nullptr;
          lo = std::min(hi, (size_t)strtoull(e + 4, &q, 10));
          if (q && *q == ':') hi = std::min(hi, (size_t)strtoull(q + 1, nullptr, 10));
        }
        fprintf(stderr, "[scratch] %p + %zu: poisoning [%zu, %zu)\n", *p, bytes, lo, hi);
      }
      if (hi > lo) GDN_HIP(hipMemsetAsync(static_cast<char *>(*p) + lo, 0xCD, hi - lo, 0));
    }
  }
  return GDN_OK;
}
namespace {
struct FenceMap {  // GDN_ALLOC_FENCE: fenced scratch pointer -> what hipMalloc returned
  std::mutex mu;
  std::map<void *, void *> base;
};
FenceMap &fence_map() {
  static FenceMap *m = new FenceMap;  // (never destroyed: frees may run during process teardown)
  return *m;
}
}  // namespace
// hipMalloc / hipFree of the arrays that are not DevBufs (graphs, gdn_dev_alloc): plain calls unless the fence is on
// (an allocation that fails while the scratch cache holds freed build blocks hands those back and tries again,
// like DevBuf::alloc and gdn_scratch_malloc do)
static hipError_t gdn_malloc_retry(void **p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    gdn_scratch_trim();
    e = hipMalloc(p, bytes);
  }
  return e;
}
hipError_t gdn_plain_malloc(void **p, size_t bytes) {
  if (!gdn_alloc_fence()) return gdn_malloc_retry(p, bytes ? bytes : 1);
  const size_t off = gdn_fence_offset(bytes ? bytes : 1);
  void *b = nullptr;
  const hipError_t e = gdn_malloc_retry(&b, (bytes ? bytes : 1) + off);
  if (e != hipSuccess) return e;
  *p = static_cast<char *>(b) + off;
  FenceMap &f = fence_map();
  std::lock_guard<std::mutex> lk(f.mu);
  f.base[*p] = b;
  return hipSuccess;
}
hipError_t gdn_plain_free(void *p) {
  if (!p) return hipSuccess;
  if (gdn_alloc_fence()) {
    FenceMap &f = fence_map();
    std::lock_guard<std::mutex> lk(f.mu);
    auto it = f.base.find(p);
    if (it != f.base.end()) {
      p = it->second;
      f.base.erase(it);
    }
  }
  return hipFree(p);
}
static int gdn_scratch_malloc_raw(void **p, size_t bytes) {
  *p = nullptr;
  if (gdn_alloc_fence()) {
    const hipError_t e = gdn_plain_malloc(p, bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      *p = nullptr;
      gdn_set_error("scratch allocation of %zu bytes -> %s", bytes, hipGetErrorString(e));
      return GDN_ERR_OOM;
    }
    return GDN_OK;
  }
  if (gdn_scratch_pooled()) {
    ScratchCache &c = scratch_cache();
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t want = scratch_round(bytes);
    if (dev >= 0 && dev < ScratchCache::kDevs) {
      {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.free_[dev].lower_bound(want);
        if (it != c.free_[dev].end() && it->first <= want + want / 4) {  // a cached block of about this size
          *p = it->second;
          c.cached[dev] -= it->first;
          c.live[*p] = std::make_pair(dev, it->first);
          c.free_[dev].erase(it);
          return GDN_OK;
        }
      }
      hipError_t e = hipMalloc(p, want);
      if (e != hipSuccess) {  // give the cache back and try once more
        (void)hipGetLastError();
        gdn_scratch_trim();
        e = hipMalloc(p, want);
      }
      if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(c.mu);
        c.live[*p] = std::make_pair(dev, want);
        return GDN_OK;
      }
      (void)hipGetLastError();
      *p = nullptr;
      gdn_set_error("scratch allocation of %zu bytes -> %s", want, hipGetErrorString(e));
      return GDN_ERR_OOM;
    }
  }
  const hipError_t e0 = hipMalloc(p, bytes ? bytes : 1);
  if (e0 != hipSuccess) {
    (void)hipGetLastError();
    *p = nullptr;
    gdn_set_error("scratch allocation of %zu bytes -> %s", bytes, hipGetErrorString(e0));
    return GDN_ERR_OOM;
  }
  return GDN_OK;
}

// The block may still be read or written by work queued on the null stream: its next user queues behind that work (every
// build runs on the null stream; a caller that used another stream synchronises it before the buffer goes out of scope).
void gdn_scratch_free(void *p) {
  if (!p) return;
  if (gdn_alloc_fence()) {
    (void)gdn_plain_free(p);
    return;
  }
  ScratchCache &c = scratch_cache();
  std::vector<void *> evict;
  int dev = -1;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.live.find(p);
    if (it != c.live.end()) {
      dev = it->second.first;
      const size_t sz = it->second.second;
      c.live.erase(it);
      c.free_[dev].insert(std::make_pair(sz, p));
      c.cached[dev] += sz;
      while (c.cached[dev] > c.keep && !c.free_[dev].empty()) {  // over the cap: the largest blocks go back
        auto last = std::prev(c.free_[dev].end());
        evict.push_back(last->second);
        c.cached[dev] -= last->first;
        c.free_[dev].erase(last);
      }
    }
  }
  if (dev < 0) {  // not from the cache (GDN_SCRATCH_POOL=0)
    (void)hipFree(p);
    return;
  }
  if (!evict.empty()) {
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    for (void *q : evict) (void)hipFree(q);
    if (cur != dev) (void)hipSetDevice(cur);
  }
}

int gdn_require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    gdn_set_error("no HIP device available (%s); libgardenia_hip has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return GDN_ERR_NO_DEVICE;
  }
  return GDN_OK;
}

// ------------------------------------------------------------------------------------------
// device-wide exclusive scan (u32 -> u64): reduce / scan-of-block-sums / down-sweep, all with
// the wave64 block scan of gdn_common.hpp.  Used for row offsets (degree -> rowptr).
// ------------------------------------------------------------------------------------------
#define SCAN_IPT 8
#define SCAN_TILE (GDN_BLOCK * SCAN_IPT)

__global__ void __launch_bounds__(GDN_BLOCK)
scan_block_sums(const uint32_t *__restrict__ in, size_t n, eoff_t *__restrict__ block_sums) {
  __shared__ eoff_t s[GDN_WAVES_PER_BLOCK];
  const size_t base = (size_t)blockIdx.x * SCAN_TILE;
  eoff_t acc = 0;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; k++) {
    const size_t i = base + (size_t)k * GDN_BLOCK + threadIdx.x;
    if (i < n) acc += in[i];
  }
  acc = gdn_block_sum(acc, s);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = acc;
}

// in-place exclusive scan of up to SCAN_TILE*? values by ONE block (loops over chunks)
__global__ void __launch_bounds__(GDN_BLOCK)
scan_single_block(eoff_t *__restrict__ data, size_t n) {
  __shared__ eoff_t s[GDN_WAVES_PER_BLOCK];
  eoff_t carry = 0;
  for (size_t base = 0; base < n; base += GDN_BLOCK) {
    const size_t i = base + threadIdx.x;
    const eoff_t v = (i < n) ? data[i] : 0;
    eoff_t total;
    const eoff_t ex = gdn_block_excl_scan(v, s, &total);
    if (i < n) data[i] = carry + ex;
    carry += total;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(GDN_BLOCK)
scan_downsweep(const uint32_t *__restrict__ in, size_t n, const eoff_t *__restrict__ block_offsets,
               eoff_t *__restrict__ out, eoff_t *__restrict__ total_out) {
  __shared__ eoff_t s[GDN_WAVES_PER_BLOCK];
  const size_t base = (size_t)blockIdx.x * SCAN_TILE;
  // thread-contiguous items so the scan order is the array order
  eoff_t v[SCAN_IPT];
  eoff_t tsum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; k++) {
    const size_t i = base + (size_t)threadIdx.x * SCAN_IPT + k;
    v[k] = (i < n) ? in[i] : 0;
    tsum += v[k];
  }
  eoff_t total;
  eoff_t ex = gdn_block_excl_scan(tsum, s, &total) + block_offsets[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_IPT; k++) {
    const size_t i = base + (size_t)threadIdx.x * SCAN_IPT + k;
    if (i < n) out[i] = ex;
    ex += v[k];
    if (i + 1 == n && total_out) *total_out = ex;  // out[n] = grand total
  }
}

// d_out has n+1 entries: out[i] = sum in[0..i), out[n] = total.
int gdn_exclusive_scan_u32_to_u64(const uint32_t *d_in, eoff_t *d_out, size_t n, hipStream_t s) {
  if (n == 0) {
    GDN_HIP(hipMemsetAsync(d_out, 0, sizeof(eoff_t), s));
    return GDN_OK;
  }
  const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  DevBuf<eoff_t> sums;
  GDN_TRY(sums.alloc_scratch(nb));
  hipLaunchKernelGGL(scan_block_sums, dim3((unsigned)nb), dim3(GDN_BLOCK), 0, s, d_in, n, sums.p);
  hipLaunchKernelGGL(scan_single_block, dim3(1), dim3(GDN_BLOCK), 0, s, sums.p, nb);
  hipLaunchKernelGGL(scan_downsweep, dim3((unsigned)nb), dim3(GDN_BLOCK), 0, s, d_in, n, sums.p, d_out,
                     d_out + n);
  GDN_HIP(hipGetLastError());
  GDN_HIP(hipStreamSynchronize(s));  // sums is freed on return
  return GDN_OK;
}

// the same scan with caller-provided scratch (ws: ceil(n / SCAN_TILE) + 1 u64), no allocation and no synchronisation:
// the layout builder runs a dozen of them back to back (a hipMalloc / hipFree pair costs 0.2 ms)
int gdn_exclusive_scan_u32_to_u64_ws(const uint32_t *d_in, eoff_t *d_out, size_t n, eoff_t *ws, hipStream_t s) {
  if (n == 0) {
    GDN_HIP(hipMemsetAsync(d_out, 0, sizeof(eoff_t), s));
    return GDN_OK;
  }
  const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(scan_block_sums, dim3((unsigned)nb), dim3(GDN_BLOCK), 0, s, d_in, n, ws);
  hipLaunchKernelGGL(scan_single_block, dim3(1), dim3(GDN_BLOCK), 0, s, ws, nb);
  hipLaunchKernelGGL(scan_downsweep, dim3((unsigned)nb), dim3(GDN_BLOCK), 0, s, d_in, n, ws, d_out, d_out + n);
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

__global__ void __launch_bounds__(GDN_BLOCK) fill_i32_kernel(int32_t *d, int32_t v, size_t n) {
  size_t i = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * GDN_BLOCK;
  for (; i < n; i += stride) d[i] = v;
}

int gdn_fill_i32(int32_t *d, int32_t v, size_t n, hipStream_t s) {
  if (n == 0) return GDN_OK;
  size_t nb = (n + GDN_BLOCK - 1) / GDN_BLOCK;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)nb), dim3(GDN_BLOCK), 0, s, d, v, n);
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

__global__ void __launch_bounds__(GDN_BLOCK)
degrees_kernel(const eoff_t *__restrict__ rowptr, int32_t m, int32_t *__restrict__ deg) {
  const int32_t v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < m) deg[v] = (int32_t)(rowptr[v + 1] - rowptr[v]);
}

__global__ void __launch_bounds__(GDN_BLOCK)
rebase_rowptr_kernel(const eoff_t *__restrict__ rowptr, int32_t row_lo, int32_t n, eoff_t *__restrict__ out) {
  const int32_t i = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (i <= n) out[i] = rowptr[row_lo + i] - rowptr[row_lo];
}

// ---- validation of caller-supplied CSR arrays (gdn_graph_upload*, gdn_graph_validate): monotone offsets inside
// [0, nnz] and column ids inside [0, n_cols).  A malformed .bin or caller array would otherwise turn into out-of-bounds
// device accesses in every solver (bitmap atomics, depth stores).  flag: bit0 offsets, bit1 column ids.
__global__ void __launch_bounds__(GDN_BLOCK)
graph_validate_kernel(const eoff_t *__restrict__ rowptr, const vid_t *__restrict__ colidx, int32_t m, uint64_t nnz,
                      int32_t n_cols, unsigned *__restrict__ flag) {
  unsigned bad = 0;
  const size_t stride = (size_t)gridDim.x * GDN_BLOCK;
  for (size_t v = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; v < (size_t)m; v += stride) {
    const eoff_t a = rowptr[v], b = rowptr[v + 1];
    if (a > b || b > nnz) bad |= 1u;
  }
  for (size_t e = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; e < nnz; e += stride) {
    const vid_t c = __builtin_nontemporal_load(colidx + e);
    if (c < 0 || c >= n_cols) bad |= 2u;
  }
  if (bad) atomicOr(flag, bad);
}

// column ids of a row shard moved into the PADDED vertex space of a sharded run (gdn_graph_slice_padded):
// vertex v of range r = [bounds[r], bounds[r+1]) -> r * chunk + (v - bounds[r])
__global__ void __launch_bounds__(GDN_BLOCK)
graph_pad_cols_kernel(vid_t *__restrict__ colidx, uint64_t nnz, const int32_t *__restrict__ bounds, int32_t world,
                      int32_t chunk) {
  __shared__ int32_t s_b[GDN_BLOCK + 1];
  for (int i = threadIdx.x; i <= world; i += GDN_BLOCK) s_b[i] = bounds[i];
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * GDN_BLOCK;
  for (size_t e = (size_t)blockIdx.x * GDN_BLOCK + threadIdx.x; e < nnz; e += stride) {
    const vid_t c = colidx[e];
    int lo = 0, hi = world;  // last r with bounds[r] <= c
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_b[mid] <= c) lo = mid;
      else hi = mid;
    }
    colidx[e] = (vid_t)((int64_t)lo * chunk + (c - s_b[lo]));
  }
}

static int graph_validate(const gdn_graph *g, int32_t n_cols, const char *who) {
  DevBuf<unsigned> flag;
  GDN_TRY(flag.alloc(1));
  GDN_HIP(hipMemset(flag.p, 0, sizeof(unsigned)));
  const uint64_t work = g->nnz > (uint64_t)g->m ? g->nnz : (uint64_t)g->m;
  unsigned nb = gdn_nblocks(work);
  if (nb > 16384u) nb = 16384u;
  hipLaunchKernelGGL(graph_validate_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, g->rowptr, g->colidx, g->m, g->nnz, n_cols, flag.p);
  unsigned f = 0;
  GDN_HIP(hipMemcpy(&f, flag.p, sizeof(unsigned), hipMemcpyDeviceToHost));
  if (f) {
    gdn_set_error("%s: malformed CSR (%s%s%s)", who, (f & 1u) ? "row offsets not ascending within [0, nnz]" : "",
                  f == 3u ? "; " : "", (f & 2u) ? "column id outside [0, n_cols)" : "");
    return GDN_ERR_INVALID;
  }
  return GDN_OK;
}

// in place: the column ids of a row shard into the padded vertex space (bounds: world + 1 host ints)
int gdn_graph_pad_cols(gdn_graph *s, int32_t world, const int32_t *bounds, int32_t chunk) {
  if (!s->nnz) return GDN_OK;
  DevBuf<int32_t> d_b;
  GDN_TRY(d_b.alloc((size_t)world + 1));
  GDN_HIP(hipMemcpy(d_b.p, bounds, ((size_t)world + 1) * 4, hipMemcpyHostToDevice));
  unsigned nb = gdn_nblocks(s->nnz);
  if (nb > 16384u) nb = 16384u;
  hipLaunchKernelGGL(graph_pad_cols_kernel, dim3(nb), dim3(GDN_BLOCK), 0, 0, s->colidx, s->nnz, d_b.p, world, chunk);
  GDN_HIP(hipDeviceSynchronize());
  return GDN_OK;
}

extern "C" {

const char *gdn_last_error(void) { return g_err; }

int gdn_option_set(const char *name, const char *value) {
  GDN_REQUIRE(name != nullptr && strncmp(name, "GDN_", 4) == 0, "option names start with GDN_");
  std::lock_guard<std::mutex> lk(g_opt_mu);
  if (value) gdn_opt_map()[name] = value;
  else gdn_opt_map().erase(name);
  return GDN_OK;
}

int gdn_option_get(const char *name, char *value, int32_t capacity) {
  GDN_REQUIRE(name != nullptr && value != nullptr && capacity > 0, "name / value");
  const char *v = gdn_option(name);
  value[0] = 0;
  if (!v) return GDN_OK;
  strncpy(value, v, (size_t)capacity - 1);
  value[capacity - 1] = 0;
  return GDN_OK;
}

int gdn_device_count(int *count) {
  GDN_REQUIRE(count != nullptr, "count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  *count = (e == hipSuccess) ? n : 0;
  return GDN_OK;
}

int gdn_set_device(int device) {
  GDN_TRY(gdn_require_device());
  GDN_HIP(hipSetDevice(device));
  return GDN_OK;
}

int gdn_dev_alloc(uint64_t bytes, void **d_ptr) {
  GDN_REQUIRE(d_ptr != nullptr, "d_ptr");
  *d_ptr = nullptr;
  GDN_TRY(gdn_require_device());
  GDN_HIP(gdn_plain_malloc(d_ptr, bytes));
  return GDN_OK;
}

int gdn_dev_free(void *d_ptr) {
  if (d_ptr) GDN_HIP(gdn_plain_free(d_ptr));
  return GDN_OK;
}

// A block set aside at the START of a process, before any build has allocated and freed memory (where hipMalloc
// puts a PageRank plan's `vals` moves its expand phase by 10-15 %, and the property belongs to the physical pages an allocation
// gets -- does memory that has never been through a build's allocations behave alike?).  One block per process; the first
// blocked PageRank plan whose `vals` fits takes it (gdn_reserve_take), its destructor frees it.
static std::mutex g_reserve_mu;
static void *g_reserve_p = nullptr;
static size_t g_reserve_bytes = 0;
int gdn_dev_reserve(uint64_t bytes) {
  GDN_TRY(gdn_require_device());
  std::lock_guard<std::mutex> lk(g_reserve_mu);
  if (g_reserve_p) {
    (void)hipFree(g_reserve_p);
    g_reserve_p = nullptr;
    g_reserve_bytes = 0;
  }
  if (bytes == 0) return GDN_OK;
  GDN_HIP(hipMalloc(&g_reserve_p, bytes));
  g_reserve_bytes = bytes;
  return GDN_OK;
}
extern "C++" void *gdn_reserve_take(size_t bytes) {
  std::lock_guard<std::mutex> lk(g_reserve_mu);
  if (!g_reserve_p || g_reserve_bytes < bytes) return nullptr;
  void *p = g_reserve_p;
  g_reserve_p = nullptr;
  g_reserve_bytes = 0;
  return p;
}

int gdn_dev_trim(uint64_t *freed_bytes) {
  if (freed_bytes) *freed_bytes = 0;
  GDN_TRY(gdn_require_device());
  uint64_t held = 0;
  {
    ScratchCache &c = scratch_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    for (int d = 0; d < ScratchCache::kDevs; d++) held += c.cached[d];
  }
  GDN_HIP(hipDeviceSynchronize());  // (a cached block may still be read by work queued on the null stream)
  gdn_scratch_trim();
  if (freed_bytes) *freed_bytes = held;
  return GDN_OK;
}

int gdn_dev_upload(void *d_dst, const void *h_src, uint64_t bytes) {
  GDN_REQUIRE(d_dst && h_src, "null pointer");
  GDN_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return GDN_OK;
}

int gdn_dev_download(void *h_dst, const void *d_src, uint64_t bytes) {
  GDN_REQUIRE(h_dst && d_src, "null pointer");
  GDN_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return GDN_OK;
}

int gdn_graph_upload(int32_t m, uint64_t nnz, const uint64_t *rowptr, const int32_t *colidx,
                     gdn_graph **out) {
  GDN_REQUIRE(out != nullptr, "out");
  *out = nullptr;
  GDN_REQUIRE(m > 0, "m must be > 0");
  GDN_REQUIRE(rowptr != nullptr, "rowptr");
  GDN_REQUIRE(colidx != nullptr || nnz == 0, "colidx");
  GDN_REQUIRE(rowptr[0] == 0 && rowptr[m] == nnz, "rowptr[0] must be 0 and rowptr[m] == nnz");
  GDN_TRY(gdn_require_device());
  gdn_graph *g = new gdn_graph();
  g->m = m;
  g->nnz = nnz;
  g->owned = true;
  hipError_t e = gdn_plain_malloc((void **)&g->rowptr, ((size_t)m + 1) * sizeof(eoff_t));
  if (e == hipSuccess) e = gdn_plain_malloc((void **)&g->colidx, (nnz ? nnz : 1) * sizeof(vid_t));
  if (e == hipSuccess) e = hipMemcpy(g->rowptr, rowptr, ((size_t)m + 1) * sizeof(eoff_t), hipMemcpyHostToDevice);
  if (e == hipSuccess && nnz) e = hipMemcpy(g->colidx, colidx, nnz * sizeof(vid_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    gdn_set_error("gdn_graph_upload: %s", hipGetErrorString(e));
    gdn_graph_free(g);
    return e == hipErrorOutOfMemory ? GDN_ERR_OOM : GDN_ERR_HIP;
  }
  const int vst = graph_validate(g, m, "gdn_graph_upload");
  if (vst != GDN_OK) {
    gdn_graph_free(g);
    return vst;
  }
  *out = g;
  return GDN_OK;
}

// rows [row_lo, row_hi) of a HOST CSR as an independent resident graph (offsets rebased on the device, column ids
// unchanged and validated against n_cols): the vertex-range shard one device of gdn_pr_multi / gdn_spmv_multi holds
int gdn_graph_upload_rows(int32_t m, const uint64_t *rowptr, const int32_t *colidx, int32_t row_lo, int32_t row_hi,
                          int32_t n_cols, gdn_graph **out) {
  GDN_REQUIRE(out != nullptr, "out");
  *out = nullptr;
  GDN_REQUIRE(rowptr != nullptr && m > 0, "rowptr / m");
  GDN_REQUIRE(0 <= row_lo && row_lo < row_hi && row_hi <= m, "row range");
  GDN_REQUIRE(rowptr[row_lo] <= rowptr[row_hi], "row offsets");
  GDN_TRY(gdn_require_device());
  const uint64_t e0 = rowptr[row_lo], nnz = rowptr[row_hi] - e0;
  GDN_REQUIRE(colidx != nullptr || nnz == 0, "colidx");
  gdn_graph *s = new gdn_graph();
  s->m = row_hi - row_lo;
  s->nnz = nnz;
  s->owned = true;
  DevBuf<eoff_t> raw;
  int st = raw.alloc((size_t)s->m + 1);
  hipError_t e = st == GDN_OK ? gdn_plain_malloc((void **)&s->rowptr, ((size_t)s->m + 1) * sizeof(eoff_t)) : hipErrorOutOfMemory;
  if (e == hipSuccess) e = gdn_plain_malloc((void **)&s->colidx, (nnz ? nnz : 1) * sizeof(vid_t));
  if (e == hipSuccess) e = hipMemcpy(raw.p, rowptr + row_lo, ((size_t)s->m + 1) * sizeof(eoff_t), hipMemcpyHostToDevice);
  if (e == hipSuccess && nnz) e = hipMemcpy(s->colidx, colidx + e0, nnz * sizeof(vid_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    gdn_set_error("gdn_graph_upload_rows: %s", hipGetErrorString(e));
    gdn_graph_free(s);
    return e == hipErrorOutOfMemory ? GDN_ERR_OOM : GDN_ERR_HIP;
  }
  hipLaunchKernelGGL(rebase_rowptr_kernel, dim3(gdn_nblocks((uint64_t)s->m + 1)), dim3(GDN_BLOCK), 0, 0, raw.p, 0, s->m, s->rowptr);
  if (hipDeviceSynchronize() != hipSuccess) {
    gdn_set_error("gdn_graph_upload_rows: rebase failed");
    gdn_graph_free(s);
    return GDN_ERR_HIP;
  }
  st = graph_validate(s, n_cols, "gdn_graph_upload_rows");
  if (st != GDN_OK) {
    gdn_graph_free(s);
    return st;
  }
  *out = s;
  return GDN_OK;
}

int gdn_graph_validate(const gdn_graph *g, int32_t n_cols) {
  GDN_REQUIRE(g != nullptr && n_cols > 0, "graph / n_cols");
  return graph_validate(g, n_cols, "gdn_graph_validate");
}

int gdn_graph_wrap_dev(int32_t m, uint64_t nnz, const uint64_t *d_rowptr, const int32_t *d_colidx,
                       gdn_graph **out) {
  GDN_REQUIRE(out != nullptr, "out");
  *out = nullptr;
  GDN_REQUIRE(m > 0 && d_rowptr != nullptr, "m / d_rowptr");
  GDN_TRY(gdn_require_device());
  gdn_graph *g = new gdn_graph();
  g->m = m;
  g->nnz = nnz;
  g->rowptr = const_cast<eoff_t *>(d_rowptr);
  g->colidx = const_cast<vid_t *>(d_colidx);
  g->owned = false;
  *out = g;
  return GDN_OK;
}

int gdn_graph_free(gdn_graph *g) {
  if (!g) return GDN_OK;
  if (g->owned) {
    if (g->rowptr) (void)gdn_plain_free(g->rowptr);
    if (g->colidx) (void)gdn_plain_free(g->colidx);
  }
  delete g;
  return GDN_OK;
}

int gdn_graph_info(const gdn_graph *g, int32_t *m, uint64_t *nnz, const uint64_t **d_rowptr,
                   const int32_t **d_colidx) {
  GDN_REQUIRE(g != nullptr, "graph");
  if (m) *m = g->m;
  if (nnz) *nnz = g->nnz;
  if (d_rowptr) *d_rowptr = g->rowptr;
  if (d_colidx) *d_colidx = g->colidx;
  return GDN_OK;
}

int gdn_graph_degrees_dev(const gdn_graph *g, int32_t *d_degree, void *stream) {
  GDN_REQUIRE(g != nullptr && d_degree != nullptr, "graph / d_degree");
  hipLaunchKernelGGL(degrees_kernel, dim3(gdn_nblocks((uint64_t)g->m)), dim3(GDN_BLOCK), 0,
                     (hipStream_t)stream, g->rowptr, g->m, d_degree);
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

int gdn_graph_slice_rows(const gdn_graph *g, int32_t row_lo, int32_t row_hi, gdn_graph **out) {
  GDN_REQUIRE(g != nullptr && out != nullptr, "graph / out");
  *out = nullptr;
  GDN_REQUIRE(0 <= row_lo && row_lo < row_hi && row_hi <= g->m, "row range");
  eoff_t ends[2];
  GDN_HIP(hipMemcpy(&ends[0], g->rowptr + row_lo, sizeof(eoff_t), hipMemcpyDeviceToHost));
  GDN_HIP(hipMemcpy(&ends[1], g->rowptr + row_hi, sizeof(eoff_t), hipMemcpyDeviceToHost));
  gdn_graph *s = new gdn_graph();
  s->m = row_hi - row_lo;
  s->nnz = ends[1] - ends[0];
  s->owned = true;
  hipError_t e = gdn_plain_malloc((void **)&s->rowptr, ((size_t)s->m + 1) * sizeof(eoff_t));
  if (e == hipSuccess) e = gdn_plain_malloc((void **)&s->colidx, (s->nnz ? s->nnz : 1) * sizeof(vid_t));
  if (e == hipSuccess && s->nnz)
    e = hipMemcpy(s->colidx, g->colidx + ends[0], s->nnz * sizeof(vid_t), hipMemcpyDeviceToDevice);
  if (e != hipSuccess) {
    gdn_set_error("gdn_graph_slice_rows: %s", hipGetErrorString(e));
    gdn_graph_free(s);
    return e == hipErrorOutOfMemory ? GDN_ERR_OOM : GDN_ERR_HIP;
  }
  hipLaunchKernelGGL(rebase_rowptr_kernel, dim3(gdn_nblocks((uint64_t)s->m + 1)), dim3(GDN_BLOCK), 0, 0,
                     g->rowptr, row_lo, s->m, s->rowptr);
  GDN_HIP(hipGetLastError());
  GDN_HIP(hipDeviceSynchronize());
  *out = s;
  return GDN_OK;
}

// smallest row b with rowptr[b] >= target (a handful of 8-byte reads of the resident offsets)
static int graph_lower_bound_row(const gdn_graph *g, eoff_t target, int32_t *row) {
  int64_t lo = 0, hi = g->m;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    eoff_t v = 0;
    GDN_HIP(hipMemcpy(&v, g->rowptr + mid, sizeof(eoff_t), hipMemcpyDeviceToHost));
    if (v >= target) hi = mid;
    else lo = mid + 1;
  }
  *row = (int32_t)lo;
  return GDN_OK;
}

int gdn_graph_balanced_ranges(const gdn_graph *g, int32_t world, int32_t *bounds) {
  GDN_REQUIRE(g != nullptr && bounds != nullptr && world >= 1, "graph / bounds / world");
  GDN_REQUIRE(world <= g->m, "more ranges than rows");
  bounds[0] = 0;
  for (int32_t r = 1; r < world; r++) {
    int32_t b = 0;
    GDN_TRY(graph_lower_bound_row(g, (eoff_t)((unsigned __int128)g->nnz * (unsigned)r / (unsigned)world), &b));
    // every range keeps at least one row (an empty shard has no plan), boundaries ascending
    if (b < bounds[r - 1] + 1) b = bounds[r - 1] + 1;
    if (b > g->m - (world - r)) b = g->m - (world - r);
    bounds[r] = b;
  }
  bounds[world] = g->m;
  return GDN_OK;
}

int gdn_graph_slice_padded(const gdn_graph *g, int32_t world, const int32_t *bounds, int32_t chunk, int32_t rank,
                           gdn_graph **out) {
  GDN_REQUIRE(g != nullptr && out != nullptr && bounds != nullptr, "graph / out / bounds");
  *out = nullptr;
  GDN_REQUIRE(world >= 1 && world <= GDN_BLOCK && rank >= 0 && rank < world, "world / rank");
  GDN_REQUIRE(bounds[0] == 0 && bounds[world] == g->m, "bounds must cover [0, m)");
  for (int32_t r = 0; r < world; r++)
    GDN_REQUIRE(bounds[r] < bounds[r + 1] && bounds[r + 1] - bounds[r] <= chunk, "every range non-empty and at most chunk rows");
  GDN_REQUIRE((int64_t)chunk * world <= 2147483647ll, "chunk * world must fit a vertex id");
  gdn_graph *s = nullptr;
  GDN_TRY(gdn_graph_slice_rows(g, bounds[rank], bounds[rank + 1], &s));
  const int st = gdn_graph_pad_cols(s, world, bounds, chunk);
  if (st != GDN_OK) {
    gdn_graph_free(s);
    return st;
  }
  *out = s;
  return GDN_OK;
}

int gdn_graph_pad_columns(gdn_graph *shard, int32_t world, const int32_t *bounds, int32_t chunk) {
  GDN_REQUIRE(shard != nullptr && bounds != nullptr, "shard / bounds");
  GDN_REQUIRE(world >= 1 && world <= GDN_BLOCK, "world");
  for (int32_t r = 0; r < world; r++)
    GDN_REQUIRE(bounds[r] < bounds[r + 1] && bounds[r + 1] - bounds[r] <= chunk, "every range non-empty and at most chunk rows");
  GDN_REQUIRE(bounds[0] == 0 && (int64_t)chunk * world <= 2147483647ll, "bounds start at 0; chunk * world must fit a vertex id");
  return gdn_graph_pad_cols(shard, world, bounds, chunk);
}

int gdn_graph_download(const gdn_graph *g, uint64_t *rowptr, int32_t *colidx) {
  GDN_REQUIRE(g != nullptr, "graph");
  if (rowptr) GDN_HIP(hipMemcpy(rowptr, g->rowptr, ((size_t)g->m + 1) * sizeof(eoff_t), hipMemcpyDeviceToHost));
  if (colidx && g->nnz) GDN_HIP(hipMemcpy(colidx, g->colidx, g->nnz * sizeof(vid_t), hipMemcpyDeviceToHost));
  return GDN_OK;
}

}  // extern "C"
