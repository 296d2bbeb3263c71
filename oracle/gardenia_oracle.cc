// gardenia_oracle.cc -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (OpenMP, raw CSR arrays: u64 row offsets, i32 column ids) of the
// reference's OpenMP solvers and serial verifiers for the six CSR hot-path kernels.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library; the product (gardenia_amd/, libgardenia_hip.so) never links or calls it.
//
// Parity status: PINNED.  Every function below is checked (tests/test_oracle.py)
//   * against the reference's golden PageRank trace test/reference/graph-pr.mtx.out:13-28,
//   * against outputs of the reference itself (oracle/_ref/ref_* binaries built from the
//     sources under /root/reference by oracle/Makefile) committed as tests/golden/*.npz,
//   * and, when oracle/_ref is present, against the reference's own verifiers on seeded
//     random graphs.
//
// Each function cites the reference file:line it follows (paths relative to the
// reference root).  Types follow include/common.h:35-47 (ScoreT/ValueT=float,
// DistT/CompT/IndexT/WeightT=int32) and include/csr_graph.h:50-51 (uint64_t offsets,
// int32 vertex ids).
//
// Build: g++ -O3 -fopenmp -ffp-contract=off -shared -fPIC (see oracle/Makefile).
// -ffp-contract=off keeps fp32 arithmetic identical to the reference's x86-64 build
// (g++ -O3 without -march: no FMA contraction).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <queue>
#include <random>
#include <unordered_map>
#include <utility>
#include <vector>
#include <omp.h>

typedef uint64_t eoff_t;
typedef int32_t vid_t;

#define ORC_MYINFINITY 1000000000           /* include/common.h:66 */
#define ORC_DIST_INF ((int32_t)(UINT_MAX / 2)) /* src/sssp/sssp.h:46 kDistInf */

extern "C" {

int orc_num_threads(void) {
  int n = 1;
#pragma omp parallel
  {
#pragma omp single
    n = omp_get_num_threads();
  }
  return n;
}

// threads of the parallel regions from here on (bench.py's N > 1 CPU leg: torch.distributed.run exports OMP_NUM_THREADS=1 to
// its ranks; rank 0 takes its share of the host's cores for the baseline instead)
void orc_set_num_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
}

// ---------------------------------------------------------------------------------
// BFS
// ---------------------------------------------------------------------------------

// Serial queue BFS: src/bfs/verifier.cc:8-27.  dist must be pre-filled with MYINFINITY
// by the caller exactly like src/bfs/main.cc:21.
void orc_bfs_serial(int32_t m, const eoff_t *rowptr, const vid_t *colidx, int32_t source,
                    int32_t *dist) {
  std::vector<int32_t> to_visit;
  to_visit.reserve(m);
  for (int32_t i = 0; i < m; i++) dist[i] = ORC_MYINFINITY;
  dist[source] = 0;
  to_visit.push_back(source);
  for (size_t it = 0; it < to_visit.size(); it++) {
    int32_t src = to_visit[it];
    for (eoff_t e = rowptr[src]; e < rowptr[src + 1]; e++) {
      vid_t dst = colidx[e];
      if (dist[dst] == ORC_MYINFINITY) {
        dist[dst] = dist[src] + 1;
        to_visit.push_back(dst);
      }
    }
  }
}

// Level-synchronous top-down BFS: src/bfs/omp_base.cc:33-64 (bfs_step :11-31).
// The reference's SlidingQueue/QueueBuffer (include/sliding_queue.h:28-120) is restated
// as a double-buffered frontier with thread-local staging and a fetch-add flush.
// Returns the number of levels ("iterations" printed at omp_base.cc:58).
int orc_bfs_topdown(int32_t m, const eoff_t *rowptr, const vid_t *colidx, int32_t source,
                    int32_t *dist) {
  int32_t *depth = dist;
#pragma omp parallel for
  for (int32_t i = 0; i < m; i++) depth[i] = ORC_MYINFINITY;
  depth[source] = 0;
  std::vector<int32_t> cur(1, source), next((size_t)m);
  size_t next_n = 0;
  int iter = 0;
  while (!cur.empty()) {
    ++iter;
    next_n = 0;
    const size_t nf = cur.size();
#pragma omp parallel
    {
      std::vector<int32_t> local;
      local.reserve(16384);
#pragma omp for
      for (size_t q = 0; q < nf; q++) {
        int32_t src = cur[q];
        for (eoff_t e = rowptr[src]; e < rowptr[src + 1]; e++) {
          vid_t dst = colidx[e];
          int32_t curr_val = depth[dst];
          if (curr_val == ORC_MYINFINITY) {
            if (__sync_bool_compare_and_swap(&depth[dst], curr_val, depth[src] + 1))
              local.push_back(dst);
          }
        }
      }
      size_t start = __sync_fetch_and_add(&next_n, local.size());
      std::copy(local.begin(), local.end(), next.begin() + start);
    }
    cur.assign(next.begin(), next.begin() + next_n);
  }
  return iter;
}

// Direction-optimising BFS: src/bfs/omp_beamer.cc:97-160 (BUStep :13-31, TDStep :35-56,
// alpha=15 beta=18 :111).  Needs the in-CSR for the bottom-up step.  Unreached vertices
// come back as MYINFINITY (:154-157).  Returns the iteration count (:150).
int orc_bfs_beamer(int32_t m, const eoff_t *out_rowptr, const vid_t *out_colidx,
                   const eoff_t *in_rowptr, const vid_t *in_colidx, int32_t source,
                   int32_t *dist) {
  const int alpha = 15, beta = 18;
  std::vector<int32_t> depths((size_t)m);
#pragma omp parallel for
  for (int32_t n = 0; n < m; n++) {
    int32_t d = (int32_t)(out_rowptr[n + 1] - out_rowptr[n]);
    depths[n] = d != 0 ? -d : -1;
  }
  int64_t scout_count = (int64_t)(out_rowptr[source + 1] - out_rowptr[source]);
  depths[source] = 0;
  std::vector<int32_t> queue(1, source), next((size_t)m);
  const size_t nwords = ((size_t)m + 63) / 64;
  std::vector<uint64_t> front(nwords, 0), curr(nwords, 0);
  int64_t edges_to_check = (int64_t)out_rowptr[m];
  int iter = 0;
  while (!queue.empty()) {
    if (scout_count > edges_to_check / alpha) {
      int64_t awake_count, old_awake_count;
      std::fill(front.begin(), front.end(), 0);
      for (size_t q = 0; q < queue.size(); q++)
        front[queue[q] >> 6] |= (uint64_t)1 << (queue[q] & 63);
      awake_count = (int64_t)queue.size();
      do {
        ++iter;
        old_awake_count = awake_count;
        awake_count = 0;
        std::fill(curr.begin(), curr.end(), 0);
#pragma omp parallel for reduction(+ : awake_count) schedule(dynamic, 1024)
        for (int32_t dst = 0; dst < m; dst++) {
          if (depths[dst] < 0) {
            for (eoff_t e = in_rowptr[dst]; e < in_rowptr[dst + 1]; e++) {
              vid_t src = in_colidx[e];
              if ((front[src >> 6] >> (src & 63)) & 1) {
                depths[dst] = depths[src] + 1;
                awake_count++;
                curr[dst >> 6] |= (uint64_t)1 << (dst & 63);
                break;
              }
            }
          }
        }
        front.swap(curr);
      } while ((awake_count >= old_awake_count) || (awake_count > m / beta));
      queue.clear();
      for (int32_t n = 0; n < m; n++)
        if ((front[n >> 6] >> (n & 63)) & 1) queue.push_back(n);
      scout_count = 1;
    } else {
      ++iter;
      edges_to_check -= scout_count;
      scout_count = 0;
      size_t next_n = 0;
      const size_t nf = queue.size();
#pragma omp parallel
      {
        std::vector<int32_t> local;
        int64_t my_scout = 0;
#pragma omp for
        for (size_t q = 0; q < nf; q++) {
          int32_t src = queue[q];
          for (eoff_t e = out_rowptr[src]; e < out_rowptr[src + 1]; e++) {
            vid_t dst = out_colidx[e];
            int32_t curr_val = depths[dst];
            if (curr_val < 0) {
              if (__sync_bool_compare_and_swap(&depths[dst], curr_val, depths[src] + 1)) {
                local.push_back(dst);
                my_scout += -curr_val;
              }
            }
          }
        }
        size_t start = __sync_fetch_and_add(&next_n, local.size());
        std::copy(local.begin(), local.end(), next.begin() + start);
        __sync_fetch_and_add(&scout_count, my_scout);
      }
      queue.assign(next.begin(), next.begin() + next_n);
    }
  }
#pragma omp parallel for
  for (int32_t i = 0; i < m; i++) dist[i] = depths[i] >= 0 ? depths[i] : ORC_MYINFINITY;
  return iter;
}

// BFSVerifier pass criterion: element-wise equality with the serial BFS,
// src/bfs/verifier.cc:31-39.  Returns 1 for "Correct", 0 for "Wrong".
int orc_bfs_verify(int32_t m, const eoff_t *rowptr, const vid_t *colidx, int32_t source,
                   const int32_t *dist_to_test) {
  std::vector<int32_t> ref((size_t)m);
  orc_bfs_serial(m, rowptr, colidx, source, ref.data());
  for (int32_t n = 0; n < m; n++)
    if (dist_to_test[n] != ref[n]) return 0;
  return 1;
}

// ---------------------------------------------------------------------------------
// PageRank
// ---------------------------------------------------------------------------------

// Pull PageRank: src/pr/omp_base.cc:8-42.  scores pre-initialised to 1/m by the caller
// (src/pr/main.cc:17-18).  Divisor is the OUT-degree, gather is over the IN-CSR
// (omp_base.cc:24-29).  trace (nullable, max_iter doubles) receives the per-iteration
// L1 error printed at omp_base.cc:35.  Returns iterations ("iter+1", :39; 100 -> 101 when
// the loop runs out, exactly like the reference's printf).
int orc_pr(int32_t m, const eoff_t *in_rowptr, const vid_t *in_colidx,
           const int32_t *out_degree, float *scores, float damping, double epsilon,
           int max_iter, double *trace) {
  const float base_score = (1.0f - damping) / m;
  float *outgoing_contrib = (float *)malloc((size_t)m * sizeof(float));
  int iter;
  for (iter = 0; iter < max_iter; iter++) {
    double error = 0;
#pragma omp parallel for
    for (int32_t n = 0; n < m; n++) outgoing_contrib[n] = scores[n] / out_degree[n];
#pragma omp parallel for reduction(+ : error) schedule(dynamic, 64)
    for (int32_t dst = 0; dst < m; dst++) {
      float incoming_total = 0;
      for (eoff_t e = in_rowptr[dst]; e < in_rowptr[dst + 1]; e++)
        incoming_total += outgoing_contrib[in_colidx[e]];
      float old_score = scores[dst];
      scores[dst] = base_score + damping * incoming_total;
      error += fabs(scores[dst] - old_score);
    }
    if (trace) trace[iter] = error;
    if (error < epsilon) break;
  }
  free(outgoing_contrib);
  return iter + 1;
}

// Fixed number of pull iterations (no convergence test); same arithmetic as orc_pr.
// Used for per-score parity at a fixed iteration count and as the cpu_baseline step.
// [row_lo,row_hi) restricts the pull loop to a row range (bounded CPU sample); contrib is
// always computed for all m vertices.  Scores outside the range are left untouched.
double orc_pr_iterate(int32_t m, const eoff_t *in_rowptr, const vid_t *in_colidx,
                      const int32_t *out_degree, float *scores, float damping, int iters,
                      int32_t row_lo, int32_t row_hi) {
  const float base_score = (1.0f - damping) / m;
  float *outgoing_contrib = (float *)malloc((size_t)m * sizeof(float));
  double error = 0;
  for (int it = 0; it < iters; it++) {
    error = 0;
#pragma omp parallel for
    for (int32_t n = 0; n < m; n++) outgoing_contrib[n] = scores[n] / out_degree[n];
#pragma omp parallel for reduction(+ : error) schedule(dynamic, 64)
    for (int32_t dst = row_lo; dst < row_hi; dst++) {
      float incoming_total = 0;
      for (eoff_t e = in_rowptr[dst]; e < in_rowptr[dst + 1]; e++)
        incoming_total += outgoing_contrib[in_colidx[e]];
      float old_score = scores[dst];
      scores[dst] = base_score + damping * incoming_total;
      error += fabs(scores[dst] - old_score);
    }
  }
  free(outgoing_contrib);
  return error;
}

// PRVerifier pass criterion: one serial PUSH iteration over the tested scores,
// src/pr/verifier.cc:40-54.  Returns the total L1 error; "Correct" iff < target_error.
double orc_pr_verify_error(int32_t m, const eoff_t *out_rowptr, const vid_t *out_colidx,
                           const float *scores_to_test, float damping) {
  const float base_score = (1.0f - damping) / m;
  float *incomming_sums = (float *)calloc((size_t)m, sizeof(float));
  double error = 0;
  for (int32_t src = 0; src < m; src++) {
    float outgoing_contrib =
        scores_to_test[src] / (int32_t)(out_rowptr[src + 1] - out_rowptr[src]);
    for (eoff_t e = out_rowptr[src]; e < out_rowptr[src + 1]; e++)
      incomming_sums[out_colidx[e]] += outgoing_contrib;
  }
  for (int32_t i = 0; i < m; i++) {
    float new_score = base_score + damping * incomming_sums[i];
    error += fabs(new_score - scores_to_test[i]);
  }
  free(incomming_sums);
  return error;
}

// ---------------------------------------------------------------------------------
// SpMV
// ---------------------------------------------------------------------------------

// y[i] += sum_k Ax[k] * x[Aj[k]]: src/spmv/omp_base.cc:22-33 (== SpmvSerial,
// src/spmv/spmv_util.h:31-42, run in parallel over rows).
void orc_spmv(int32_t m, const eoff_t *Ap, const vid_t *Aj, const float *Ax, const float *x,
              float *y) {
#pragma omp parallel for schedule(dynamic, 1024)
  for (int32_t i = 0; i < m; i++) {
    float sum = y[i];
    for (eoff_t jj = Ap[i]; jj < Ap[i + 1]; jj++) sum += x[Aj[jj]] * Ax[jj];
    y[i] = sum;
  }
}

// maximum_relative_error: src/spmv/spmv_util.h:16-29.  SpmvVerifier passes iff the
// result is <= 5*sqrt(FLT_EPSILON) (src/spmv/verifier.cc:24).
float orc_spmv_max_rel_error(const float *A, const float *B, int64_t N) {
  float max_error = 0;
  float eps = std::sqrt(std::numeric_limits<float>::epsilon());
  for (int64_t i = 0; i < N; i++) {
    const float a = A[i], b = B[i];
    const float error = std::abs(a - b);
    if (error != 0) max_error = std::max(max_error, error / (std::abs(a) + std::abs(b) + eps));
  }
  return max_error;
}

// bytes_per_spmv: src/spmv/spmv_util.h:6-13 (the reference's own byte model, 4-byte
// IndexT row pointers).
uint64_t orc_bytes_per_spmv(int64_t m, int64_t nnz) {
  return 2 * 4 * (uint64_t)m + 4 * (uint64_t)nnz + 2 * 4 * (uint64_t)nnz + 2 * 4 * (uint64_t)m;
}

// ---------------------------------------------------------------------------------
// SSSP
// ---------------------------------------------------------------------------------

// Serial Dijkstra: src/sssp/verifier.cc:8-39.  Unreached = kDistInf.
void orc_sssp_dijkstra(int32_t m, const eoff_t *rowptr, const vid_t *colidx,
                       const int32_t *weight, int32_t source, int32_t *dist) {
  for (int32_t i = 0; i < m; i++) dist[i] = ORC_DIST_INF;
  typedef std::pair<int32_t, int32_t> WN;
  std::priority_queue<WN, std::vector<WN>, std::greater<WN> > mq;
  dist[source] = 0;
  mq.push(std::make_pair(0, source));
  while (!mq.empty()) {
    int32_t td = mq.top().first;
    int32_t src = mq.top().second;
    mq.pop();
    if (td == dist[src]) {
      for (eoff_t e = rowptr[src]; e < rowptr[src + 1]; e++) {
        vid_t dst = colidx[e];
        int32_t wt = weight[e];
        if (td + wt < dist[dst]) {
          dist[dst] = td + wt;
          mq.push(std::make_pair(td + wt, dst));
        }
      }
    }
  }
}

// Delta-stepping: src/sssp/omp_base.cc:12-97 (relax loop :38-65, bin vote :66-72, bin
// copy :80-87).  dist pre-filled with kDistInf by the caller (src/sssp/main.cc:25).
void orc_sssp_delta(int32_t m, const eoff_t *rowptr, const vid_t *colidx,
                    const int32_t *weight, int32_t source, int32_t delta, int32_t *dist) {
  const size_t kInf = (size_t)(UINT_MAX / 2);
  for (int32_t i = 0; i < m; i++) dist[i] = ORC_DIST_INF;
  dist[source] = 0;
  size_t nnz = (size_t)rowptr[m];
  int32_t *frontier = (int32_t *)malloc((nnz > 0 ? nnz : 1) * sizeof(int32_t));
  size_t shared_indexes[2] = {0, kInf};
  size_t frontier_tails[2] = {1, 0};
  frontier[0] = source;
#pragma omp parallel
  {
    std::vector<std::vector<int32_t> > local_bins(0);
    int iter = 0;
    while ((int32_t)shared_indexes[iter & 1] != ORC_DIST_INF) {
      size_t &curr_bin_index = shared_indexes[iter & 1];
      size_t &next_bin_index = shared_indexes[(iter + 1) & 1];
      size_t &curr_frontier_tail = frontier_tails[iter & 1];
      size_t &next_frontier_tail = frontier_tails[(iter + 1) & 1];
#pragma omp for nowait schedule(dynamic, 64)
      for (size_t i = 0; i < curr_frontier_tail; i++) {
        int32_t src = frontier[i];
        if (dist[src] >= delta * (int32_t)curr_bin_index) {
          for (eoff_t e = rowptr[src]; e < rowptr[src + 1]; e++) {
            vid_t dst = colidx[e];
            int32_t old_dist = dist[dst];
            int32_t new_dist = dist[src] + weight[e];
            if (new_dist < old_dist) {
              bool changed_dist = true;
              while (!__sync_bool_compare_and_swap(&dist[dst], old_dist, new_dist)) {
                old_dist = dist[dst];
                if (old_dist <= new_dist) {
                  changed_dist = false;
                  break;
                }
              }
              if (changed_dist) {
                size_t dest_bin = (size_t)(new_dist / delta);
                if (dest_bin >= local_bins.size()) local_bins.resize(dest_bin + 1);
                local_bins[dest_bin].push_back(dst);
              }
            }
          }
        }
      }
      for (size_t i = curr_bin_index; i < local_bins.size(); i++) {
        if (!local_bins[i].empty()) {
#pragma omp critical
          next_bin_index = std::min(next_bin_index, i);
          break;
        }
      }
#pragma omp barrier
#pragma omp single nowait
      {
        curr_bin_index = kInf;
        curr_frontier_tail = 0;
      }
      if (next_bin_index < local_bins.size()) {
        size_t copy_start =
            __sync_fetch_and_add(&next_frontier_tail, local_bins[next_bin_index].size());
        std::copy(local_bins[next_bin_index].begin(), local_bins[next_bin_index].end(),
                  frontier + copy_start);
        local_bins[next_bin_index].resize(0);
      }
      iter++;
#pragma omp barrier
    }
  }
  free(frontier);
}

// SSSPVerifier pass criterion: exact equality with Dijkstra, src/sssp/verifier.cc:41-49.
int orc_sssp_verify(int32_t m, const eoff_t *rowptr, const vid_t *colidx, const int32_t *weight,
                    int32_t source, const int32_t *dist_to_test) {
  std::vector<int32_t> ref((size_t)m);
  orc_sssp_dijkstra(m, rowptr, colidx, weight, source, ref.data());
  for (int32_t n = 0; n < m; n++)
    if (dist_to_test[n] != ref[n]) return 0;
  return 1;
}

// ---------------------------------------------------------------------------------
// Connected components
// ---------------------------------------------------------------------------------

// Shiloach-Vishkin: src/cc/omp_base.cc:6-50 (hook :24-37, shortcut :38-43).  The hook
// always points the higher root at the lower label, so the fixpoint label of every vertex
// is the minimum vertex id of its component.  Returns the round count (:47).
int orc_cc_sv(int32_t m, const eoff_t *rowptr, const vid_t *colidx, int32_t *comp) {
#pragma omp parallel for
  for (int32_t n = 0; n < m; n++) comp[n] = n;
  bool change = true;
  int iter = 0;
  while (change) {
    change = false;
    iter++;
#pragma omp parallel for schedule(dynamic, 64)
    for (int32_t src = 0; src < m; src++) {
      int32_t comp_src = comp[src];
      for (eoff_t e = rowptr[src]; e < rowptr[src + 1]; e++) {
        int32_t comp_dst = comp[colidx[e]];
        if (comp_src == comp_dst) continue;
        int32_t high_comp = comp_src > comp_dst ? comp_src : comp_dst;
        int32_t low_comp = comp_src + (comp_dst - high_comp);
        if (high_comp == comp[high_comp]) {
          change = true;
          comp[high_comp] = low_comp;
        }
      }
    }
#pragma omp parallel for
    for (int32_t n = 0; n < m; n++) {
      while (comp[n] != comp[comp[n]]) comp[n] = comp[comp[n]];
    }
  }
  return iter;
}

// Link: src/cc/omp_afforest.cc:12-25.
static void orc_link(int32_t u, int32_t v, int32_t *comp) {
  int32_t p1 = comp[u];
  int32_t p2 = comp[v];
  while (p1 != p2) {
    int32_t high = p1 > p2 ? p1 : p2;
    int32_t low = p1 + (p2 - high);
    int32_t p_high = comp[high];
    if ((p_high == low) ||
        (p_high == high && __sync_bool_compare_and_swap(&comp[high], high, low)))
      break;
    p1 = comp[comp[high]];
    p2 = comp[low];
  }
}

// Compress: src/cc/omp_afforest.cc:28-35.
static void orc_compress(int32_t m, int32_t *comp) {
#pragma omp parallel for schedule(static, 2048)
  for (int32_t n = 0; n < m; n++) {
    while (comp[n] != comp[comp[n]]) comp[n] = comp[comp[n]];
  }
}

// SampleFrequentElement: src/cc/verifier.cc:13-33 (1024 samples, default-seeded
// std::mt19937, std::uniform_int_distribution).
int32_t orc_sample_frequent_element(int32_t m, const int32_t *comp, int64_t num_samples) {
  std::unordered_map<int32_t, int> sample_counts(32);
  typedef std::unordered_map<int32_t, int>::value_type kvp_type;
  std::mt19937 gen;
  std::uniform_int_distribution<int32_t> distribution(0, m - 1);
  for (int64_t i = 0; i < num_samples; i++) {
    int32_t n = distribution(gen);
    sample_counts[comp[n]]++;
  }
  auto most_frequent =
      std::max_element(sample_counts.begin(), sample_counts.end(),
                       [](const kvp_type &a, const kvp_type &b) { return a.second < b.second; });
  return most_frequent->first;
}

// Afforest: src/cc/omp_afforest.cc:37-83, neighbor_rounds = 2.  in_rowptr == NULL means
// an undirected (symmetrized) graph (:56-63); otherwise the directed branch (:64-76).
void orc_cc_afforest(int32_t m, const eoff_t *rowptr, const vid_t *colidx,
                     const eoff_t *in_rowptr, const vid_t *in_colidx, int32_t *comp) {
  const int32_t neighbor_rounds = 2;
#pragma omp parallel for
  for (int32_t n = 0; n < m; n++) comp[n] = n;
  for (int32_t r = 0; r < neighbor_rounds; ++r) {
#pragma omp parallel for
    for (int32_t src = 0; src < m; src++) {
      eoff_t b = rowptr[src], e = rowptr[src + 1];
      eoff_t off = std::min<eoff_t>((eoff_t)r, e - b);  // csr_graph.h:275-280 out_neigh(v, start)
      if (b + off < e) orc_link(src, colidx[b + off], comp);
    }
    orc_compress(m, comp);
  }
  int32_t c = orc_sample_frequent_element(m, comp, 1024);
#pragma omp parallel for schedule(dynamic, 2048)
  for (int32_t u = 0; u < m; u++) {
    if (comp[u] == c) continue;
    eoff_t b = rowptr[u], e = rowptr[u + 1];
    eoff_t off = std::min<eoff_t>((eoff_t)neighbor_rounds, e - b);
    for (eoff_t k = b + off; k < e; k++) orc_link(u, colidx[k], comp);
    if (in_rowptr) {
      for (eoff_t k = in_rowptr[u]; k < in_rowptr[u + 1]; k++) orc_link(u, in_colidx[k], comp);
    }
  }
  orc_compress(m, comp);
}

// CCVerifier pass criterion: src/cc/verifier.cc:62-124 -- every label class must be closed
// under the (out-)edges and a per-label BFS must visit every vertex.  It checks the
// partition, not the label values.  Returns 1 for "Correct".
int orc_cc_verify(int32_t m, const eoff_t *rowptr, const vid_t *colidx, const int32_t *comp_test) {
  std::map<int32_t, int32_t> label_to_source;
  std::vector<char> visited((size_t)m, 0);
  std::vector<int32_t> frontier;
  for (int32_t i = 0; i < m; i++) label_to_source[comp_test[i]] = i;
  frontier.reserve(m);
  for (auto it = label_to_source.begin(); it != label_to_source.end(); ++it) {
    int32_t curr_label = it->first;
    int32_t source = it->second;
    frontier.clear();
    frontier.push_back(source);
    visited[source] = 1;
    for (size_t q = 0; q < frontier.size(); q++) {
      int32_t src = frontier[q];
      for (eoff_t e = rowptr[src]; e < rowptr[src + 1]; e++) {
        vid_t dst = colidx[e];
        if (comp_test[dst] != curr_label) return 0;
        if (!visited[dst]) {
          visited[dst] = 1;
          frontier.push_back(dst);
        }
      }
    }
  }
  for (int32_t n = 0; n < m; n++)
    if (!visited[n]) return 0;
  return 1;
}

// ---------------------------------------------------------------------------------
// Triangle counting
// ---------------------------------------------------------------------------------

// DAG orientation: src/common/graph.cc:67-113 (keep u->v iff deg[v] > deg[u] or equal
// degree and v > u, :80-81).  Pass new_colidx == NULL to only count: new_rowptr (m+1)
// is always filled.  Returns the oriented edge count.
uint64_t orc_tc_orient(int32_t m, const eoff_t *rowptr, const vid_t *colidx, eoff_t *new_rowptr,
                       vid_t *new_colidx) {
  new_rowptr[0] = 0;
  for (int32_t src = 0; src < m; src++) {
    eoff_t dsrc = rowptr[src + 1] - rowptr[src];
    eoff_t cnt = 0;
    for (eoff_t e = rowptr[src]; e < rowptr[src + 1]; e++) {
      vid_t dst = colidx[e];
      eoff_t ddst = rowptr[dst + 1] - rowptr[dst];
      if (ddst > dsrc || (ddst == dsrc && dst > src)) {
        if (new_colidx) new_colidx[new_rowptr[src] + cnt] = dst;
        cnt++;
      }
    }
    new_rowptr[src + 1] = new_rowptr[src] + cnt;
  }
  return new_rowptr[m];
}

// Merge-count of two ascending lists: include/VertexSet.h:65-76 (get_intersect_num).
static inline uint64_t orc_intersect_num(const vid_t *a, eoff_t na, const vid_t *b, eoff_t nb) {
  uint64_t num = 0;
  eoff_t il = 0, ir = 0;
  while (il < na && ir < nb) {
    vid_t left = a[il], right = b[ir];
    if (left <= right) il++;
    if (right <= left) ir++;
    if (left == right) num++;
  }
  return num;
}

// TCSolver on an (already oriented) CSR: src/tc/omp_base.cc:6-26 == TCVerifier
// src/tc/verifier.cc:8-24 run in parallel.
uint64_t orc_tc(int32_t m, const eoff_t *rowptr, const vid_t *colidx) {
  uint64_t counter = 0;
#pragma omp parallel for reduction(+ : counter) schedule(dynamic, 64)
  for (int32_t u = 0; u < m; u++) {
    const vid_t *yu = colidx + rowptr[u];
    eoff_t nu = rowptr[u + 1] - rowptr[u];
    for (eoff_t k = 0; k < nu; k++) {
      vid_t v = yu[k];
      counter += orc_intersect_num(yu, nu, colidx + rowptr[v], rowptr[v + 1] - rowptr[v]);
    }
  }
  return counter;
}


// ---------------------------------------------------------------------------------
// Betweenness centrality, one source
// ---------------------------------------------------------------------------------

// Brandes from ONE source like src/bc/omp_base.cc:55-105 (num_iters = 1): forward BFS with int path counts
// (PBFS :16-53: a neighbour found at depth+1 receives the path count of its parent), then the dependencies from the
// deepest level back (:78-93: delta_src = SUM over successors of pc[src]/pc[dst] * (1 + delta[dst]), float, in CSR
// order), scores[src] += delta_src, finally every score divided by the largest (:95-101).  scores is in/out like
// src/bc/main.cc:21 (the caller zero-fills).  depths_out (nullable, m ints, -1 = unreached) and
// path_counts_out (nullable) expose the forward phase for tests.  Returns the number of BFS levels.
// The forward phase here is level-synchronous and serial per level, so path counts come out in the same (wrapping
// int) arithmetic as the reference's atomics; the per-source sums keep the reference's CSR order.
int orc_bc(int32_t m, const eoff_t *rowptr, const vid_t *colidx, int32_t source, float *scores, int32_t *depths_out,
           int32_t *path_counts_out) {
  std::vector<int32_t> depths((size_t)m, -1), order;
  std::vector<uint32_t> pc((size_t)m, 0u);  // the reference adds ints; unsigned keeps the wrap defined
  std::vector<size_t> level_ptr;
  order.reserve((size_t)m);
  depths[source] = 0;
  pc[source] = 1;
  order.push_back(source);
  level_ptr.push_back(0);
  size_t head = 0;
  int depth = 0;
  while (head < order.size()) {
    const size_t tail = order.size();
    level_ptr.push_back(tail);
    depth++;
    for (size_t i = head; i < tail; i++) {
      const vid_t src = order[i];
      for (eoff_t k = rowptr[src]; k < rowptr[src + 1]; k++) {
        const vid_t dst = colidx[k];
        if (depths[dst] == -1) {
          depths[dst] = depth;
          order.push_back(dst);
        }
        if (depths[dst] == depth) pc[dst] += pc[src];
      }
    }
    head = tail;
  }
  // level_ptr = {0, end of level 0, end of level 1, ...}; the last entry repeats the end (empty level)
  std::vector<float> deltas((size_t)m, 0.0f);
  const int nlev = (int)level_ptr.size() - 1;  // levels 0 .. nlev-1 (the last one may be empty)
  for (int d = nlev - 1; d >= 0; d--) {
#pragma omp parallel for schedule(dynamic, 64)
    for (long long i = (long long)level_ptr[d]; i < (long long)level_ptr[d + 1]; i++) {
      const vid_t src = order[(size_t)i];
      float delta_src = 0.0f;
      for (eoff_t k = rowptr[src]; k < rowptr[src + 1]; k++) {
        const vid_t dst = colidx[k];
        if (depths[dst] == depths[src] + 1)
          delta_src += static_cast<float>((int32_t)pc[src]) / static_cast<float>((int32_t)pc[dst]) * (1 + deltas[dst]);
      }
      deltas[src] = delta_src;
      scores[src] += delta_src;
    }
  }
  float biggest = 0.0f;
  for (int32_t n = 0; n < m; n++) biggest = std::max(biggest, scores[n]);
  for (int32_t n = 0; n < m; n++) scores[n] = scores[n] / biggest;
  if (depths_out) memcpy(depths_out, depths.data(), (size_t)m * 4);
  if (path_counts_out) memcpy(path_counts_out, pc.data(), (size_t)m * 4);
  int levels = 0;
  for (int d = 0; d < nlev; d++)
    if (level_ptr[d + 1] > level_ptr[d]) levels = d + 1;
  return levels;
}

// Verifier criterion of src/bc/verifier.cc:17-44,128-132: serial Brandes (orc_bc IS serial in its forward phase and
// sums in the same order as the verifier's :107-119), then |a - b| <= 1e-4 * (|a| + |b|) + 1e-4 for every vertex.
// Returns 1 = "Correct", 0 = "POSSIBLE FAILURE".
int orc_bc_verify(int32_t m, const eoff_t *rowptr, const vid_t *colidx, int32_t source, const float *scores_to_test) {
  std::vector<float> want((size_t)m, 0.0f);
  orc_bc(m, rowptr, colidx, source, want.data(), nullptr, nullptr);
  for (int32_t i = 0; i < m; i++) {
    const double a = scores_to_test[i], b = want[i];
    if (std::isnan(a) != std::isnan(b)) return 0;
    if (std::isnan(a)) continue;  // 0/0 on both sides (no vertex has a dependency)
    if (fabs(a - b) > 1e-4 * (fabs(a) + fabs(b)) + 1e-4) return 0;
  }
  return 1;
}
// Delta PageRank, src/pr/omp_delta.cc:52-107 (CUDA twin src/pr/delta.cu:140-202): scores start at 1/m (caller), deltas at
// 1/m; an iteration PULLS the deltas of ALL vertices over the in-CSR (omp_delta.cc:32-49) while the frontier holds at
// least m / push_div vertices, and PUSHES the deltas of the frontier's vertices over the out-CSR otherwise (:12-28;
// push_div = 10 at omp_delta.cc:69, 8 at delta.cu:178); then delta = d * sum (first iteration: base + d * sum - 1/m,
// :84-89), score += delta, and a vertex enters the next frontier iff |delta| > epsilon2 * score (:92, epsilon2 = 1e-3,
// pr.h:8).  Stops when the frontier is empty, after max_iter iterations or when sum |delta| < epsilon (:96-103).
// The push is serial here, in ascending vertex order (the reference's order depends on the thread schedule).
// Returns the iterations executed (the reference PRINTS iter + 1, omp_delta.cc:105).
// trace_diff[it], trace_items[it] (frontier after the iteration), trace_mode[it] (0 pull, 1 push): nullable.
int orc_pr_delta(int32_t m, const eoff_t *in_rowptr, const vid_t *in_colidx, const eoff_t *out_rowptr,
                 const vid_t *out_colidx, const int32_t *out_degree, float *scores, float damping, double epsilon,
                 float epsilon2, int max_iter, int push_div, double *trace_diff, int32_t *trace_items,
                 int32_t *trace_mode) {
  const float base_score = (1.0f - damping) / m;
  const float init_score = 1.0f / m;
  std::vector<float> sums((size_t)m, 0.0f), deltas((size_t)m, init_score), contrib((size_t)m, 0.0f);
  std::vector<vid_t> queue((size_t)m), next;
  for (int32_t i = 0; i < m; i++) queue[i] = i;
  int iter = 0;
  while (!queue.empty() && iter < max_iter) {
    ++iter;
    const bool push = (int64_t)queue.size() < (int64_t)(m / push_div);
    if (push) {
      for (vid_t src : queue) {
        const eoff_t b = out_rowptr[src], e = out_rowptr[src + 1];
        const int degree = (int)(e - b);
        const float c = deltas[src] / (float)degree;
        for (eoff_t k = b; k < e; k++) sums[out_colidx[k]] += c;
      }
    } else {
#pragma omp parallel for
      for (int32_t n = 0; n < m; n++) contrib[n] = deltas[n] / out_degree[n];
#pragma omp parallel for schedule(dynamic, 64)
      for (int32_t dst = 0; dst < m; dst++) {
        float incoming_total = 0;
        for (eoff_t k = in_rowptr[dst]; k < in_rowptr[dst + 1]; k++) incoming_total += contrib[in_colidx[k]];
        sums[dst] = incoming_total;
      }
    }
    next.clear();
    double error = 0;
    for (int32_t u = 0; u < m; u++) {
      if (iter == 1) {
        deltas[u] = base_score + damping * sums[u];
        deltas[u] -= init_score;
      } else {
        deltas[u] = damping * sums[u];
      }
      scores[u] += deltas[u];
      sums[u] = 0;
      if (fabs(deltas[u]) > epsilon2 * scores[u]) next.push_back(u);
      error += fabs(deltas[u]);
    }
    queue.swap(next);
    if (trace_diff) trace_diff[iter - 1] = error;
    if (trace_items) trace_items[iter - 1] = (int32_t)queue.size();
    if (trace_mode) trace_mode[iter - 1] = push ? 1 : 0;
    if (error < epsilon) break;
  }
  return iter;
}
}  // extern "C"
