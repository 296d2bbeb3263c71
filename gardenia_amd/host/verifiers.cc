// verifiers.cc -- the harness' serial checkers, restating the PASS CRITERIA of the reference's
// verifier.cc files (they are part of the harness, not of the compute path):
//   BFS  exact compare with a serial queue BFS            src/bfs/verifier.cc:8-40
//   PR   one serial PUSH iteration, total L1 < target     src/pr/verifier.cc:40-54
//   SpMV max relative error <= 5*sqrt(FLT_EPSILON)        src/spmv/verifier.cc:7-28, spmv_util.h:16-29
//   SSSP exact compare with serial Dijkstra               src/sssp/verifier.cc:8-50
//   CC   every label class closed under edges + covered   src/cc/verifier.cc:62-124
//   TC   serial merge-intersect recount on the DAG        src/tc/verifier.cc:8-24
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <map>
#include <queue>

#include "gardenia_host.hpp"

bool BFSVerifier(Graph &g, int source, DistT *depth_to_test) {
  printf("Verifying...\n");
  const VertexId m = g.V();
  std::vector<DistT> depth(m, MYINFINITY);
  std::vector<int> to_visit;
  to_visit.reserve(m);
  depth[source] = 0;
  to_visit.push_back(source);
  for (size_t it = 0; it < to_visit.size(); it++) {
    int src = to_visit[it];
    for (VertexId dst : g.N(src))
      if (depth[dst] == MYINFINITY) {
        depth[dst] = depth[src] + 1;
        to_visit.push_back(dst);
      }
  }
  bool ok = true;
  for (VertexId n = 0; n < m; n++) ok &= depth_to_test[n] == depth[n];
  printf(ok ? "Correct\n" : "Wrong\n");
  return ok;
}

bool PRVerifier(Graph &g, ScoreT *scores_to_test, double target_error) {
  printf("Verifying...\n");
  const VertexId m = g.V();
  const ScoreT base_score = (1.0f - kDamp) / m;
  std::vector<ScoreT> sums(m, 0);
  double error = 0;
  for (VertexId src = 0; src < m; src++) {
    ScoreT c = scores_to_test[src] / g.get_degree(src);
    for (VertexId dst : g.out_neigh(src)) sums[dst] += c;
  }
  for (VertexId i = 0; i < m; i++) error += fabs(base_score + kDamp * sums[i] - scores_to_test[i]);
  if (error < target_error) printf("Correct\n");
  else printf("Total Error: %f\n", error);
  return error < target_error;
}

bool SpmvVerifier(Graph &g, const ValueT *Ax, const ValueT *x, ValueT *y0, ValueT *test_y) {
  printf("Verifying...\n");
  const VertexId m = g.V();
  const uint64_t *Ap = g.in_rowptr();
  const VertexId *Aj = g.in_colidx();
  ValueT max_error = 0, eps = std::sqrt(std::numeric_limits<ValueT>::epsilon());
  for (VertexId i = 0; i < m; i++) {
    ValueT sum = y0[i];
    for (uint64_t jj = Ap[i]; jj < Ap[i + 1]; jj++) sum += x[Aj[jj]] * Ax[jj];
    const ValueT err = std::abs(test_y[i] - sum);
    if (err != 0) max_error = std::max(max_error, err / (std::abs(test_y[i]) + std::abs(sum) + eps));
  }
  printf("\t[max error %9f]\n", max_error);
  const bool ok = !(max_error > 5 * eps);
  printf(ok ? "Correct\n" : "POSSIBLE FAILURE\n");
  return ok;
}

bool SSSPVerifier(Graph &g, int source, DistT *weight, DistT *dist_to_test) {
  printf("Verifying...\n");
  std::vector<DistT> d(g.V(), (DistT)kDistInf);
  typedef std::pair<DistT, IndexT> WN;
  std::priority_queue<WN, std::vector<WN>, std::greater<WN> > mq;
  d[source] = 0;
  mq.push(std::make_pair(0, source));
  while (!mq.empty()) {
    DistT td = mq.top().first;
    IndexT src = mq.top().second;
    mq.pop();
    if (td != d[src]) continue;
    uint64_t off = g.edge_begin(src);
    for (VertexId dst : g.N(src)) {
      DistT wt = weight[off++];
      if (td + wt < d[dst]) {
        d[dst] = td + wt;
        mq.push(std::make_pair(td + wt, dst));
      }
    }
  }
  bool ok = true;
  for (VertexId n = 0; n < g.V(); n++) ok &= dist_to_test[n] == d[n];
  printf(ok ? "Correct\n" : "Wrong\n");
  return ok;
}

bool CCVerifier(Graph &g, CompT *comp_test) {
  printf("Verifying...\n");
  const VertexId m = g.V();
  std::map<int, int> label_to_source;
  std::vector<char> visited(m, 0);
  std::vector<int> frontier;
  for (VertexId i = 0; i < m; i++) label_to_source[comp_test[i]] = i;
  frontier.reserve(m);
  for (auto &kv : label_to_source) {
    frontier.clear();
    frontier.push_back(kv.second);
    visited[kv.second] = 1;
    for (size_t q = 0; q < frontier.size(); q++)
      for (VertexId dst : g.N(frontier[q])) {
        if (comp_test[dst] != kv.first) {
          printf("Wrong\n");
          return false;
        }
        if (!visited[dst]) {
          visited[dst] = 1;
          frontier.push_back(dst);
        }
      }
  }
  for (VertexId n = 0; n < m; n++)
    if (!visited[n]) {
      printf("Wrong\n");
      return false;
    }
  printf("Correct\n");
  return true;
}

bool TCVerifier(Graph &g, uint64_t test_total) {
  printf("Verifying...\n");
  g.orientation();  // the reference orients while loading (src/tc/main.cc:12); the host copy is only used here
  uint64_t total = 0;
  for (VertexId u = 0; u < g.V(); u++) {
    VertexSet yu = g.N(u);
    for (VertexId v : yu) total += (uint64_t)yu.get_intersect_num(g.N(v));
  }
  printf(total == test_total ? "Correct\n" : "Wrong\n");
  printf("total %llu test_total %llu\n", (unsigned long long)total, (unsigned long long)test_total);
  return total == test_total;
}

// src/bc/verifier.cc:69-146: serial Brandes from the same source (depths and int path counts by a queue BFS, the
// dependencies from the deepest level back with the successors regenerated from the depths), scores normalised to the
// largest, then |a - b| <= 1e-4 * (|a| + |b|) + 1e-4 per vertex (:24-30); a NaN on both sides (0/0: nothing lies between)
// counts as equal.
bool BCVerifier(Graph &g, int source, int num_iters, ScoreT *scores_to_test) {
  printf("Verifying...\n");
  const VertexId m = g.V();
  std::vector<ScoreT> scores(m, 0);
  int max_depth = 0;
  for (int iter = 0; iter < num_iters; iter++) {
    std::vector<int> depths(m, -1), path_counts(m, 0);
    std::vector<VertexId> to_visit;
    to_visit.reserve(m);
    depths[source] = 0;
    path_counts[source] = 1;
    to_visit.push_back(source);
    for (size_t it = 0; it < to_visit.size(); it++) {
      const VertexId src = to_visit[it];
      for (VertexId dst : g.N(src)) {
        if (depths[dst] == -1) {
          depths[dst] = depths[src] + 1;
          to_visit.push_back(dst);
        }
        if (depths[dst] == depths[src] + 1) path_counts[dst] = (int)((unsigned)path_counts[dst] + (unsigned)path_counts[src]);
      }
    }
    std::vector<std::vector<VertexId>> at_depth;
    for (VertexId n = 0; n < m; n++)
      if (depths[n] != -1) {
        if (depths[n] >= (int)at_depth.size()) at_depth.resize(depths[n] + 1);
        at_depth[depths[n]].push_back(n);
      }
    max_depth = (int)at_depth.size();
    std::vector<ScoreT> deltas(m, 0);
    for (int d = max_depth - 1; d >= 0; d--)
      for (VertexId src : at_depth[d]) {
        ScoreT delta_src = 0;
        for (VertexId dst : g.N(src))
          if (depths[dst] == depths[src] + 1)
            delta_src += static_cast<ScoreT>(path_counts[src]) / static_cast<ScoreT>(path_counts[dst]) * (1 + deltas[dst]);
        deltas[src] = delta_src;
        scores[src] += delta_src;
      }
  }
  ScoreT biggest = 0;
  for (VertexId n = 0; n < m; n++) biggest = scores[n] > biggest ? scores[n] : biggest;
  for (VertexId n = 0; n < m; n++) scores[n] = scores[n] / biggest;
  printf("\titerations = %d.\n", max_depth);
  printf("\tmax_score = %.6f.\n", biggest);
  bool ok = true;
  for (VertexId n = 0; n < m && ok; n++) {
    const double a = scores_to_test[n], b = scores[n];
    if ((a != a) != (b != b)) ok = false;
    else if (a == a && std::fabs(a - b) > 1e-4 * (std::fabs(a) + std::fabs(b)) + 1e-4) ok = false;
    if (!ok) printf("score_test[%d] (%f) != score[%d] (%f)\n", (int)n, scores_to_test[n], (int)n, scores[n]);
  }
  printf(ok ? "Correct\n" : "POSSIBLE FAILURE\n");
  return ok;
}
