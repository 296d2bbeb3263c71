// solvers.cc -- the reference's XxxSolver(Graph&, ...) signatures, each ONE call into the C-ABI.
// Prints the lines the reference solvers print ("\titerations = %d.", "\truntime [..] = %f ms.")
// so outputs stay diff-able (SURVEY 5 metrics/logging).  No CPU fallback: failures throw.
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "gardenia_host.hpp"

static void must(int status, const char *what) {
  if (status != GDN_OK) throw std::runtime_error(std::string(what) + ": " + gdn_last_error());
}

void BFSSolver(Graph &g, int source, DistT *dist) {  // src/bfs/bfs.h:43
  gdn_stats st;
  const bool rev = g.has_reverse_graph();
  must(gdn_bfs(g.V(), g.E(), g.out_rowptr(), g.out_colidx(), rev ? g.in_rowptr() : nullptr,
               rev ? g.in_colidx() : nullptr, source, dist, &st), "BFSSolver");
  printf("\titerations = %d.\n", st.iterations);
  printf("\truntime [hip_gfx950] = %f ms.\n", st.solve_ms);
  printf("\tthroughput = %f GTEPS (%llu edges)\n", st.edges_traversed / (st.solve_ms * 1e-3) / 1e9,
         (unsigned long long)st.edges_traversed);
}

// GDN_NUM_GPUS: the devices PRSolver / SpmvSolver spread over -- the multi-GPU analogue of the OMP_NUM_THREADS the
// reference's OpenMP solvers honour (src/pr/omp_base.cc:11-16 prints the thread count the same way)
static int num_gpus() {
  const char *e = getenv("GDN_NUM_GPUS");
  const int n = e ? atoi(e) : 1;
  return n < 1 ? 1 : n;
}

void PRSolver(Graph &g, ScoreT *scores) {  // src/pr/pr.h:31
  std::vector<int32_t> deg(g.V());
  for (VertexId v = 0; v < g.V(); v++) deg[v] = g.get_degree(v);
  gdn_stats st;
  const int ngpus = num_gpus();
  if (ngpus > 1) {
    printf("Launching HIP PR solver (%d GPUs) ...\n", ngpus);
    must(gdn_pr_multi(g.V(), g.E(), g.in_rowptr(), g.in_colidx(), deg.data(), scores, kDamp, EPSILON, MAX_ITER, ngpus, nullptr,
                      &st), "PRSolver");
  } else {
    must(gdn_pr(g.V(), g.E(), g.in_rowptr(), g.in_colidx(), deg.data(), scores, kDamp, EPSILON, MAX_ITER, &st), "PRSolver");
  }
  // the per-iteration trace exactly as src/pr/omp_base.cc:35 prints it (iteration numbers from 1)
  std::vector<double> trace(MAX_ITER);
  int32_t n = 0;
  must(gdn_pr_last_trace(MAX_ITER, &n, trace.data()), "PRSolver");
  for (int32_t i = 0; i < n && i < MAX_ITER; i++) printf(" %2d    %lf\n", i + 1, trace[i]);
  printf("\titerations = %d.\n", st.iterations);
  printf("\truntime [hip_gfx950] = %f ms.\n", st.solve_ms);
}

// The delta variant of PRSolver (src/pr/delta.cu:140; link-time choice in the reference, src/pr/Makefile): prints the
// reference's per-iteration "pull:" / "push:" lines.
void PRDeltaSolver(Graph &g, ScoreT *scores) {
  gdn_graph *gi = nullptr, *go = nullptr;
  gdn_pr_delta_plan *plan = nullptr;
  void *d_scores = nullptr;
  const uint64_t bytes = sizeof(ScoreT) * (uint64_t)g.V();
  gdn_stats st;
  try {
    must(gdn_graph_upload(g.V(), g.E(), g.in_rowptr(), g.in_colidx(), &gi), "PRDeltaSolver");
    must(gdn_graph_upload(g.V(), g.E(), g.out_rowptr(), g.out_colidx(), &go), "PRDeltaSolver");
    must(gdn_pr_delta_plan_create(gi, go, GDN_LAYOUT_AUTO, &plan), "PRDeltaSolver");
    must(gdn_dev_alloc(bytes, &d_scores), "PRDeltaSolver");
    must(gdn_dev_upload(d_scores, scores, bytes), "PRDeltaSolver");
    must(gdn_pr_delta_run(plan, (float *)d_scores, kDamp, EPSILON, epsilon2, MAX_ITER, 8, &st), "PRDeltaSolver");
    must(gdn_dev_download(scores, d_scores, bytes), "PRDeltaSolver");
    std::vector<double> diff(MAX_ITER);
    std::vector<int32_t> mode(MAX_ITER);
    int32_t n = 0;
    must(gdn_pr_delta_trace(plan, MAX_ITER, &n, diff.data(), nullptr, mode.data()), "PRDeltaSolver");
    for (int32_t i = 0; i < n; i++) printf("%s %2d    %lf\n", (mode[i] & 1) ? "push:" : "pull:", i + 1, diff[i]);
  } catch (...) {
    if (d_scores) gdn_dev_free(d_scores);
    if (plan) gdn_pr_delta_plan_free(plan);
    if (go) gdn_graph_free(go);
    if (gi) gdn_graph_free(gi);
    throw;
  }
  gdn_dev_free(d_scores);
  gdn_pr_delta_plan_free(plan);
  gdn_graph_free(go);
  gdn_graph_free(gi);
  printf("\titerations = %d.\n", st.iterations);
  printf("\truntime [hip_gfx950_delta] = %f ms.\n", st.solve_ms);
}

void SpmvSolver(Graph &g, const ValueT *Ax, const ValueT *x, ValueT *y) {  // src/spmv/spmv.h:29
  gdn_stats st;
  const int ngpus = num_gpus();
  if (ngpus > 1)
    must(gdn_spmv_multi(g.V(), g.E(), g.in_rowptr(), g.in_colidx(), Ax, x, y, ngpus, nullptr, &st), "SpmvSolver");
  else
    must(gdn_spmv(g.V(), g.E(), g.in_rowptr(), g.in_colidx(), Ax, x, y, &st), "SpmvSolver");
  const double t = st.solve_ms;
  const double bytes = 8.0 * (g.V() + 1) + 12.0 * g.E() + 8.0 * g.V();  // SURVEY 8d byte model
  printf("\truntime [hip_gfx950] = %.4f ms ( %5.2f GFLOP/s %5.1f GB/s)\n", t, t > 0 ? 2.0 * g.E() / t / 1e6 : 0.0,
         t > 0 ? bytes / t / 1e6 : 0.0);
}

void SSSPSolver(Graph &g, int source, DistT *weight, DistT *dist, int delta) {  // src/sssp/sssp.h:47
  gdn_stats st;
  must(gdn_sssp(g.V(), g.E(), g.out_rowptr(), g.out_colidx(), weight, source, delta, dist, &st), "SSSPSolver");
  printf("\titerations = %d.\n", st.iterations);
  printf("\truntime [hip_gfx950] = %f ms.\n", st.solve_ms);
}

void CCSolver(Graph &g, CompT *comp) {  // src/cc/cc.h:28
  gdn_stats st;
  // reverse graph for directed inputs, the graph itself for symmetrized ones (in_rowptr() aliases)
  const bool rev = g.has_reverse_graph();
  must(gdn_cc(g.V(), g.E(), g.out_rowptr(), g.out_colidx(), rev ? g.in_rowptr() : nullptr, rev ? g.in_colidx() : nullptr,
              comp, &st), "CCSolver");
  printf("iterations = %d\n", st.iterations);
  printf("runtime [hip_gfx950] = %f seconds\n", st.solve_ms * 1e-3);
}

void TCSolver(Graph &g, uint64_t &total) {  // src/tc/tc.h:7
  gdn_stats st;
  must(gdn_tc(g.V(), g.E(), g.out_rowptr(), g.out_colidx(), /*oriented=*/0, &total, &st), "TCSolver");
  printf("runtime [hip_gfx950] = %f sec (orientation %f sec)\n", st.solve_ms * 1e-3, st.prep_ms * 1e-3);
  printf("throughput = %f billion Traversed Edges Per Second (TEPS)\n", st.edges_traversed / (st.solve_ms * 1e-3) / 1e9);
}

void BCSolver(Graph &g, int source, ScoreT *scores) {  // src/bc/bc.h:37
  gdn_stats st;
  must(gdn_bc(g.V(), g.E(), g.out_rowptr(), g.out_colidx(), source, scores, &st), "BCSolver");
  printf("\titerations = %d.\n", st.iterations);
  printf("\truntime [hip_gfx950] = %f ms.\n", st.solve_ms);
}
