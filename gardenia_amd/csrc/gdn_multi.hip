// gdn_multi.hip -- the multi-GPU drop-in solvers: gdn_pr_multi / gdn_spmv_multi (`gdn_pr(..., ngpus, ...)`,
// SURVEY 8e).
//
// The reference has no multi-GPU hot path (its vestige: the edge-list slicing stub of include/graph_gpu.h:145-165);
// a GARDENIA maintainer who links PRSolver (src/pr/pr.h:31) / SpmvSolver (src/spmv/spmv.h:29) reaches N devices through
// these two entries.  ONE process, ONE host thread per device (HIP's current device is per thread), the rows of the
// in-CSR cut into N contiguous vertex ranges of about nnz/N edges each (binary search on the row offsets).
//
// Vertex space: range r = [bounds[r], bounds[r+1]) is moved to the slot [r*chunk, r*chunk + len_r) of a PADDED space of
// chunk*N ids (chunk = longest range, rounded to 4): every device's slice of the replicated contribution vector then is
// an equal-sized all-gather slot although the ranges hold different numbers of rows.
//
// PageRank iteration on device r: the fused pull of its rows (gdn_pr_pull_rows_dev: reads the full contribution vector,
// writes its slice of the next one), then the exchange of the slices:
//   * "rccl" (default on distinct devices): in-place ncclAllGather of the chunk-sized slots, one communicator per
//     device (ncclCommInitAll), called from the device's thread on its compute stream.  librccl is dlopen'ed on first
//     use: a single-GPU process never loads it.
//   * "p2p" (GDN_MULTI_EXCHANGE=p2p, and whenever a device appears twice in `devices` -- RCCL refuses that; this is how
//     a 1-GPU box exercises the path): every device copies its slice into its peers' vectors with hipMemcpyPeerAsync
//     on a copy stream, PIPELINED in row-range parts behind the pull kernels (part j is sent while part j+1 is computed).
// The 8-byte L1 change of every device is read back per iteration (the convergence test needs it on the host, like
// src/pr/base.cu:124) and summed in rank order: deterministic.  With the PB layout the scores are bit-identical to the
// single-device solver's whatever N is (integer accumulation is order independent).
#include <dlfcn.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "gdn_common.hpp"

// ---- the slice of the RCCL API used here (rccl/rccl.h:36,236,260,466,678), resolved with dlsym
namespace {
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;  // ncclSuccess == 0
enum { kNcclFloat = 7 };   // ncclFloat32, rccl.h:466
struct Rccl {
  void *h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
  bool load() {
    if (h) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) {
      why = dlerror() ? dlerror() : "dlopen(librccl) failed";
      return false;
    }
    CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
    CommAbort = (decltype(CommAbort))dlsym(h, "ncclCommAbort");
    AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
    GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !AllGather) {
      why = "librccl lacks ncclCommInitAll / ncclCommDestroy / ncclAllGather";
      h = nullptr;
      return false;
    }
    return true;
  }
};
Rccl g_rccl;
std::mutex g_rccl_mu;

struct Barrier {
  std::mutex mu;
  std::condition_variable cv;
  int n, waiting = 0;
  unsigned long gen = 0;
  explicit Barrier(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    const unsigned long g = gen;
    if (++waiting == n) {
      waiting = 0;
      gen++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g; });
    }
  }
};

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// nnz-balanced vertex ranges on the HOST offsets (binary search on row_offsets); every range keeps a row
void balanced_bounds(int32_t m, const uint64_t *rowptr, int n, std::vector<int32_t> &b) {
  b.assign((size_t)n + 1, 0);
  const uint64_t nnz = rowptr[m];
  for (int r = 1; r < n; r++) {
    const uint64_t target = (uint64_t)((unsigned __int128)nnz * (unsigned)r / (unsigned)n);
    int64_t lo = 0, hi = m;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (rowptr[mid] >= target) hi = mid;
      else lo = mid + 1;
    }
    int32_t v = (int32_t)lo;
    if (v < b[r - 1] + 1) v = b[r - 1] + 1;
    if (v > m - (n - r)) v = m - (n - r);
    b[r] = v;
  }
  b[n] = m;
}

struct Shared {
  int n = 0;
  std::vector<int> dev;
  std::vector<int32_t> bounds;
  int32_t chunk = 0;
  bool use_rccl = false;
  std::vector<ncclComm_t> comm;
  std::mutex abort_mu;
  bool aborted = false;  // the communicators were ended with ncclCommAbort (they must not be destroyed again)
  Barrier *bar = nullptr;
  std::vector<float *> contrib[2];  // per rank: its two replicated vectors (padded space)
  std::vector<double> diff[2];      // per rank L1 change, slot = iteration & 1
  std::vector<int> rc;
  std::vector<std::string> err;
  std::vector<int32_t> n_bins;
  std::vector<double> t_h2d, t_prep;
  double solve_ms = 0;
  int iterations = 0;
  double last = 0;
  std::vector<double> trace;
  int failed() const {
    for (int r = 0; r < n; r++)
      if (rc[r] != GDN_OK) return rc[r];
    return GDN_OK;
  }
};

#define MT_CHECK(call)                         \
  do {                                         \
    if (rc == GDN_OK) {                        \
      rc = (call);                             \
      if (rc != GDN_OK) S.err[r] = gdn_last_error(); \
    }                                          \
  } while (0)
#define MT_HIP(call)                                                             \
  do {                                                                           \
    if (rc == GDN_OK) {                                                          \
      hipError_t e__ = (call);                                                   \
      if (e__ != hipSuccess) {                                                   \
        rc = e__ == hipErrorOutOfMemory ? GDN_ERR_OOM : GDN_ERR_HIP;             \
        S.err[r] = std::string(#call) + " -> " + hipGetErrorString(e__);         \
      }                                                                          \
    }                                                                            \
  } while (0)

// every rank leaves a collective phase together: publish rc, barrier, read everybody's, barrier (nobody publishes the
// next phase's status before everyone has read this one's -- all ranks take the same branch)
#define MT_SYNC_OR_QUIT()        \
  do {                           \
    S.rc[r] = rc;                \
    S.bar->wait();               \
    const int f__ = S.failed();  \
    S.bar->wait();               \
    if (f__) goto done;          \
  } while (0)

void pr_rank(Shared &S, int r, int32_t m, const uint64_t *in_rowptr, const int32_t *in_colidx, const int32_t *out_degree,
             float *scores, float damping, double epsilon, int32_t max_iter) {
  int rc = GDN_OK;
  const int n = S.n;
  const int32_t lo = S.bounds[r], hi = S.bounds[r + 1], ml = hi - lo, chunk = S.chunk;
  const int32_t m_pad = chunk * n, base = r * chunk;
  gdn_graph *shard = nullptr;
  gdn_pr_plan *plan = nullptr;
  DevBuf<int32_t> d_deg;
  DevBuf<float> d_scores, d_c[2];
  DevBuf<double> d_diff;
  hipStream_t copy = nullptr;
  std::vector<hipEvent_t> ev;
  int parts = 1;
  double t0 = now_ms();
  if (hipSetDevice(S.dev[r]) != hipSuccess) {
    rc = GDN_ERR_NO_DEVICE;
    S.err[r] = "hipSetDevice failed";
  }
  MT_CHECK(gdn_graph_upload_rows(m, in_rowptr, in_colidx, lo, hi, m, &shard));
  MT_CHECK(gdn_graph_pad_cols(shard, n, S.bounds.data(), chunk));
  MT_CHECK(d_deg.alloc((size_t)ml));
  MT_CHECK(d_scores.alloc((size_t)ml));
  MT_CHECK(d_c[0].alloc((size_t)m_pad + 4));
  MT_CHECK(d_c[1].alloc((size_t)m_pad + 4));
  MT_CHECK(d_diff.alloc(1));
  MT_HIP(hipMemcpy(d_deg.p, out_degree + lo, (size_t)ml * 4, hipMemcpyHostToDevice));
  MT_HIP(hipMemcpy(d_scores.p, scores + lo, (size_t)ml * 4, hipMemcpyHostToDevice));
  MT_HIP(hipMemset(d_c[0].p, 0, ((size_t)m_pad + 4) * 4));
  MT_HIP(hipMemset(d_c[1].p, 0, ((size_t)m_pad + 4) * 4));
  S.t_h2d[r] = now_ms() - t0;
  t0 = now_ms();
  MT_CHECK(gdn_pr_plan_create(shard, d_deg.p, m_pad, base, GDN_LAYOUT_AUTO, &plan));
  MT_CHECK(gdn_pr_plan_set_base(plan, m));
  if (rc == GDN_OK) {
    int32_t nb = 0;
    (void)gdn_pr_plan_bins(plan, &nb);
    S.n_bins[r] = nb;
  }
  if (!S.use_rccl) MT_HIP(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
  S.t_prep[r] = now_ms() - t0;
  S.contrib[0][r] = d_c[0].p;
  S.contrib[1][r] = d_c[1].p;
  MT_SYNC_OR_QUIT();
  {
    // pipeline parts of the p2p exchange: every part's accumulate launch keeps about 200 workgroups or more
    // (the same count on every rank); the RCCL all-gather moves whole slots
    if (!S.use_rccl) {
      int32_t nbmin = S.n_bins[0];
      for (int q = 1; q < n; q++) nbmin = S.n_bins[q] < nbmin ? S.n_bins[q] : nbmin;
      // (round 6: the parts are ranges of ONE launch -- gdn_pr_pull_parts_dev --; four of them wherever a part still holds half
      // a round of workgroups: the exchange, not the pull, is what a step waits for)
      parts = nbmin >= 256 ? (nbmin / 128 > 4 ? 4 : nbmin / 128) : 1;
      if (const char *e = gdn_test_option("GDN_MULTI_PARTS")) parts = atoi(e) >= 1 && atoi(e) <= 8 ? atoi(e) : parts;  // (test hook)
      ev.resize((size_t)parts);
      for (int j = 0; j < parts; j++) MT_HIP(hipEventCreateWithFlags(&ev[(size_t)j], hipEventDisableTiming));
    }
    int32_t seg = (chunk + parts - 1) / parts;
    seg = (seg + 3) & ~3;
    auto exchange = [&](float *vec, int32_t r0, int32_t r1, hipStream_t s) {  // rows [r0,r1) of this rank's slot to the peers
      const int32_t a = r0 < ml ? r0 : ml, b = r1 < ml ? r1 : ml;
      if (b <= a) return;
      const int which = vec == d_c[0].p ? 0 : 1;
      for (int q = 0; q < n && rc == GDN_OK; q++) {
        if (q == r) continue;
        MT_HIP(hipMemcpyPeerAsync(S.contrib[which][q] + base + a, S.dev[q], vec + base + a, S.dev[r], (size_t)(b - a) * 4, s));
      }
    };
    // contrib = score / out_degree of the own rows (src/pr/base.cu:14), then the first exchange
    MT_CHECK(gdn_pr_contrib_dev(plan, d_scores.p, d_c[0].p, nullptr));
    // a rank that skips a collective its peers have entered leaves them waiting for ever: every rank publishes its
    // status and all agree BEFORE each all-gather; a failed enqueue aborts every communicator so that the peers'
    // kernels end
    auto all_gather = [&](float *vec) {
      if (g_rccl.AllGather(vec + base, vec, (size_t)chunk, kNcclFloat, S.comm[r], nullptr) != 0) {
        rc = GDN_ERR_HIP;
        S.err[r] = "ncclAllGather failed";
        std::lock_guard<std::mutex> lk(S.abort_mu);
        if (g_rccl.CommAbort && !S.aborted) {
          S.aborted = true;
          for (int q = 0; q < n; q++)
            if (S.comm[q]) (void)g_rccl.CommAbort(S.comm[q]);
        }
      }
    };
    if (S.use_rccl) {
      MT_SYNC_OR_QUIT();
      all_gather(d_c[0].p);
    } else {
      exchange(d_c[0].p, 0, chunk, nullptr);
    }
    MT_HIP(hipDeviceSynchronize());
    MT_SYNC_OR_QUIT();
    const double t_solve = now_ms();
    int cur = 0, iter = 0;
    double total = 0;
    for (iter = 0; iter < max_iter; iter++) {
      float *cin = d_c[cur].p, *cout = d_c[cur ^ 1].p;
      if (S.use_rccl) {
        MT_CHECK(gdn_pr_pull_dev(plan, cin, d_scores.p, cout, d_diff.p, damping, nullptr));
        MT_SYNC_OR_QUIT();
        all_gather(cout);
      } else {
        // ONE launch per phase whose rows become final part by part (tickets); the copies of part j are queued on the copy
        // stream behind the one-wave kernel that waits for part j's tickets, and overlap the accumulation of the later parts
        int32_t ends[8];
        for (int j = 0; j < parts; j++) {
          const int32_t r1 = (j == parts - 1) ? chunk : ((j + 1) * seg < chunk ? (j + 1) * seg : chunk);
          ends[j] = r1 < ml ? r1 : ml;
        }
        MT_CHECK(gdn_pr_pull_parts_dev(plan, cin, d_scores.p, cout, d_diff.p, damping, parts, ends, nullptr));
        for (int j = 0; j < parts && rc == GDN_OK; j++) {
          const int32_t r0 = j * seg < chunk ? j * seg : chunk, r1 = (j == parts - 1) ? chunk : ((j + 1) * seg < chunk ? (j + 1) * seg : chunk);
          MT_CHECK(gdn_pr_wait_part_dev(plan, j, copy));
          exchange(cout, r0, r1, copy);
        }
      }
      double h = 0;
      MT_HIP(hipMemcpy(&h, d_diff.p, sizeof(double), hipMemcpyDeviceToHost));  // blocks on the null stream: pull done
      if (copy) MT_HIP(hipStreamSynchronize(copy));
      S.diff[iter & 1][r] = h;
      MT_SYNC_OR_QUIT();  // every slice has landed everywhere, every L1 change is published
      total = 0;
      for (int q = 0; q < n; q++) total += S.diff[iter & 1][q];
      if (r == 0) S.trace.push_back(total);
      cur ^= 1;
      if (total < epsilon) break;  // src/pr/omp_base.cc:36
    }
    if (r == 0) {
      S.solve_ms = now_ms() - t_solve;
      S.iterations = iter + 1;  // the reference prints iter + 1 (omp_base.cc:39: MAX_ITER + 1 when it did not converge), like gdn_pr
      S.last = total;
    }
    MT_CHECK(gdn_pr_plan_check(plan));
    MT_HIP(hipMemcpy(scores + lo, d_scores.p, (size_t)ml * 4, hipMemcpyDeviceToHost));
    S.rc[r] = rc;
  }
done:
  if (rc != GDN_OK) S.rc[r] = rc;
  for (hipEvent_t e : ev)
    if (e) (void)hipEventDestroy(e);
  if (copy) (void)hipStreamDestroy(copy);
  gdn_pr_plan_free(plan);
  gdn_graph_free(shard);
}

void spmv_rank(Shared &S, int r, int32_t m, const uint64_t *Ap, const int32_t *Aj, const float *Ax, const float *x, float *y,
               double *solve_ms) {
  int rc = GDN_OK;
  const int32_t lo = S.bounds[r], hi = S.bounds[r + 1], ml = hi - lo;
  gdn_graph *shard = nullptr;
  gdn_spmv_plan *plan = nullptr;
  DevBuf<float> d_Ax, d_x, d_y;
  double t0 = now_ms();
  if (hipSetDevice(S.dev[r]) != hipSuccess) {
    rc = GDN_ERR_NO_DEVICE;
    S.err[r] = "hipSetDevice failed";
  }
  const uint64_t e0 = Ap[lo], nnz = Ap[hi] - e0;
  MT_CHECK(gdn_graph_upload_rows(m, Ap, Aj, lo, hi, m, &shard));
  MT_CHECK(d_Ax.alloc((size_t)nnz));
  MT_CHECK(d_x.alloc((size_t)m));
  MT_CHECK(d_y.alloc((size_t)ml));
  if (nnz) MT_HIP(hipMemcpy(d_Ax.p, Ax + e0, (size_t)nnz * 4, hipMemcpyHostToDevice));
  MT_HIP(hipMemcpy(d_x.p, x, (size_t)m * 4, hipMemcpyHostToDevice));  // x is replicated: a one-shot multiply needs no exchange
  MT_HIP(hipMemcpy(d_y.p, y + lo, (size_t)ml * 4, hipMemcpyHostToDevice));
  S.t_h2d[r] = now_ms() - t0;
  t0 = now_ms();
  // one multiply: the merge-path layout needs no build (see gdn_spmv)
  MT_CHECK(gdn_spmv_plan_create_cols(shard, nullptr, m, GDN_LAYOUT_CSR, &plan));
  S.t_prep[r] = now_ms() - t0;
  S.rc[r] = rc;
  S.bar->wait();
  const int failed = S.failed();
  S.bar->wait();
  if (!failed) {
    t0 = now_ms();
    MT_CHECK(gdn_spmv_dev(plan, d_Ax.p, d_x.p, d_y.p, nullptr));
    MT_HIP(hipDeviceSynchronize());
    S.rc[r] = rc;
    S.bar->wait();
    if (r == 0) *solve_ms = now_ms() - t0;
    MT_HIP(hipMemcpy(y + lo, d_y.p, (size_t)ml * 4, hipMemcpyDeviceToHost));
  }
  if (rc != GDN_OK) S.rc[r] = rc;
  gdn_spmv_plan_free(plan);
  gdn_graph_free(shard);
}

int multi_setup(Shared &S, int32_t m, const uint64_t *rowptr, int32_t ngpus, const int32_t *devices, bool want_exchange) {
  int ndev = 0;
  GDN_TRY(gdn_require_device());
  GDN_HIP(hipGetDeviceCount(&ndev));
  int n = ngpus;
  if (n > m) n = m;  // a range without a row has no shard
  S.n = n;
  S.dev.resize((size_t)n);
  bool dup = false;
  // GDN_MULTI_DEVICES=0,0 (with devices == NULL): the rank -> device map of a harness that cannot pass one (the
  // XxxSolver(Graph&, ...) wrappers; two ranks on one device is how a 1-GPU box runs the path)
  std::vector<int32_t> envdev;
  if (!devices)
    if (const char *e = gdn_option("GDN_MULTI_DEVICES")) {
      for (const char *c = e; *c;) {
        char *end = nullptr;
        const long v = strtol(c, &end, 10);
        if (end == c || envdev.size() >= 64 || (*end && *end != ',' && *end != ' ')) {
          gdn_set_error("GDN_MULTI_DEVICES=\"%s\": expected at most 64 device numbers separated by commas", e);
          return GDN_ERR_INVALID;
        }
        envdev.push_back((int32_t)v);
        c = end;
        while (*c == ',' || *c == ' ') c++;
      }
      if ((int)envdev.size() >= n) devices = envdev.data();
    }
  for (int r = 0; r < n; r++) {
    S.dev[r] = devices ? devices[r] : r;
    if (S.dev[r] < 0 || S.dev[r] >= ndev) {
      gdn_set_error("gdn_*_multi: device %d of rank %d does not exist (%d HIP devices; pass `devices` to map ranks)", S.dev[r], r, ndev);
      return GDN_ERR_INVALID;
    }
    for (int q = 0; q < r; q++) dup |= S.dev[q] == S.dev[r];
  }
  balanced_bounds(m, rowptr, n, S.bounds);
  int32_t longest = 0;
  for (int r = 0; r < n; r++) longest = S.bounds[r + 1] - S.bounds[r] > longest ? S.bounds[r + 1] - S.bounds[r] : longest;
  S.chunk = (longest + 3) & ~3;
  if ((int64_t)S.chunk * n > 2147483647ll) {
    gdn_set_error("gdn_*_multi: the padded vertex space (%d x %d) does not fit a vertex id", S.chunk, n);
    return GDN_ERR_INVALID;
  }
  for (int k = 0; k < 2; k++) {
    S.contrib[k].assign((size_t)n, nullptr);
    S.diff[k].assign((size_t)n, 0.0);
  }
  S.rc.assign((size_t)n, GDN_OK);
  S.err.assign((size_t)n, std::string());
  S.n_bins.assign((size_t)n, 0);
  S.t_h2d.assign((size_t)n, 0.0);
  S.t_prep.assign((size_t)n, 0.0);
  S.use_rccl = false;
  const char *ex_env = gdn_option("GDN_MULTI_EXCHANGE");
  // GDN_MULTI_EXCHANGE=rccl takes the RCCL path even for one rank (a 1-GPU box then drives librccl: dlopen, communicator,
  // the in-place all-gather call)
  if (want_exchange && (n > 1 || (ex_env && ex_env[0] == 'r'))) {
    const char *e = ex_env;
    const bool want_p2p = dup || (e && e[0] == 'p');
    if (!want_p2p) {
      std::lock_guard<std::mutex> lk(g_rccl_mu);
      if (!g_rccl.load()) {
        if (e && e[0] == 'r') {
          gdn_set_error("gdn_pr_multi: GDN_MULTI_EXCHANGE=rccl but librccl could not be loaded: %s", g_rccl.why.c_str());
          return GDN_ERR_HIP;
        }
        fprintf(stderr, "[gardenia_hip] librccl not loadable (%s): contrib slices go by hipMemcpyPeerAsync\n", g_rccl.why.c_str());
      } else {
        S.comm.assign((size_t)n, nullptr);
        const ncclResult_t st = g_rccl.CommInitAll(S.comm.data(), n, S.dev.data());
        if (st != 0) {
          gdn_set_error("ncclCommInitAll(%d devices): %s", n, g_rccl.GetErrorString ? g_rccl.GetErrorString(st) : "error");
          return GDN_ERR_HIP;
        }
        S.use_rccl = true;
      }
    }
  }
  return GDN_OK;
}

int multi_finish(Shared &S, const char *who) {
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (size_t r = 0; r < S.comm.size(); r++)
    if (S.comm[r] && !S.aborted) (void)g_rccl.CommDestroy(S.comm[r]);
  for (int r = 0; r < S.n; r++)
    if (S.rc[r] != GDN_OK) {
      gdn_set_error("%s: rank %d (device %d): %s", who, r, S.dev[r], S.err[r].empty() ? "failed" : S.err[r].c_str());
      return S.rc[r];
    }
  return GDN_OK;
}
}  // namespace

extern "C" {

int gdn_pr_multi(int32_t m, uint64_t nnz, const uint64_t *in_rowptr, const int32_t *in_colidx, const int32_t *out_degree,
                 float *scores, float damping, double epsilon, int32_t max_iter, int32_t ngpus, const int32_t *devices,
                 gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && in_rowptr && out_degree && scores && (in_colidx || nnz == 0), "null argument");
  GDN_REQUIRE(max_iter >= 1, "max_iter");
  GDN_REQUIRE(ngpus >= 1 && ngpus <= 64, "ngpus");
  GDN_REQUIRE(in_rowptr[0] == 0 && in_rowptr[m] == nnz, "rowptr[0] must be 0 and rowptr[m] == nnz");
  if (ngpus == 1 && !devices) {  // the single-device solver (squished state, trace included)
    return gdn_pr(m, nnz, in_rowptr, in_colidx, out_degree, scores, damping, epsilon, max_iter, stats);
  }
  Shared S;
  GDN_TRY(multi_setup(S, m, in_rowptr, ngpus, devices, true));
  Barrier bar(S.n);
  S.bar = &bar;
  int home = 0;
  (void)hipGetDevice(&home);
  std::vector<std::thread> th;
  for (int r = 0; r < S.n; r++)
    th.emplace_back(pr_rank, std::ref(S), r, m, in_rowptr, in_colidx, out_degree, scores, damping, epsilon, max_iter);
  for (auto &t : th) t.join();
  (void)hipSetDevice(home);
  const int rc = multi_finish(S, "gdn_pr_multi");
  gdn_pr_trace_set(S.trace.data(), (int32_t)S.trace.size());
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->iterations = S.iterations;
    stats->reserved = S.use_rccl ? 1 : 2;  // exchange that ran: 1 = RCCL all-gather, 2 = peer copies
    stats->solve_ms = S.solve_ms;
    for (int r = 0; r < S.n; r++) {
      stats->h2d_ms = S.t_h2d[r] > stats->h2d_ms ? S.t_h2d[r] : stats->h2d_ms;
      stats->prep_ms = S.t_prep[r] > stats->prep_ms ? S.t_prep[r] : stats->prep_ms;
    }
    stats->last_error = S.last;
    stats->edges_traversed = nnz * (uint64_t)S.iterations;
  }
  return rc;
}

int gdn_spmv_multi(int32_t m, uint64_t nnz, const uint64_t *Ap, const int32_t *Aj, const float *Ax, const float *x, float *y,
                   int32_t ngpus, const int32_t *devices, gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && Ap && x && y && ((Ax && Aj) || nnz == 0), "null argument");
  GDN_REQUIRE(ngpus >= 1 && ngpus <= 64, "ngpus");
  GDN_REQUIRE(Ap[0] == 0 && Ap[m] == nnz, "Ap[0] must be 0 and Ap[m] == nnz");
  if (ngpus == 1 && !devices) return gdn_spmv(m, nnz, Ap, Aj, Ax, x, y, stats);
  Shared S;
  GDN_TRY(multi_setup(S, m, Ap, ngpus, devices, false));
  Barrier bar(S.n);
  S.bar = &bar;
  int home = 0;
  (void)hipGetDevice(&home);
  double solve_ms = 0;
  std::vector<std::thread> th;
  for (int r = 0; r < S.n; r++) th.emplace_back(spmv_rank, std::ref(S), r, m, Ap, Aj, Ax, x, y, &solve_ms);
  for (auto &t : th) t.join();
  (void)hipSetDevice(home);
  const int rc = multi_finish(S, "gdn_spmv_multi");
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->iterations = 1;
    stats->solve_ms = solve_ms;
    for (int r = 0; r < S.n; r++) {
      stats->h2d_ms = S.t_h2d[r] > stats->h2d_ms ? S.t_h2d[r] : stats->h2d_ms;
      stats->prep_ms = S.t_prep[r] > stats->prep_ms ? S.t_prep[r] : stats->prep_ms;
    }
    stats->edges_traversed = nnz;
  }
  return rc;
}

/* the vertex ranges gdn_pr_multi / gdn_spmv_multi would cut a host CSR into (bounds: ngpus + 1 entries) */
int gdn_multi_ranges(int32_t m, const uint64_t *rowptr, int32_t ngpus, int32_t *bounds, int32_t *chunk) {
  GDN_REQUIRE(m > 0 && rowptr && bounds && ngpus >= 1 && ngpus <= m, "null argument / ngpus");
  std::vector<int32_t> b;
  balanced_bounds(m, rowptr, ngpus, b);
  int32_t longest = 0;
  for (int r = 0; r < ngpus; r++) {
    bounds[r] = b[(size_t)r];
    longest = b[(size_t)r + 1] - b[(size_t)r] > longest ? b[(size_t)r + 1] - b[(size_t)r] : longest;
  }
  bounds[ngpus] = m;
  if (chunk) *chunk = (longest + 3) & ~3;
  return GDN_OK;
}

}  // extern "C"
