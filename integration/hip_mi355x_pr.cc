// src/pr/hip_mi355x.cc -- PRSolver (src/pr/pr.h:31, called by src/pr/main.cc:19) on MI355X.
#include <vector>

#include "pr.h"
#include "gdn_binding.h"

void PRSolver(Graph &g, ScoreT *scores) {
  const VertexId m = g.V();
  std::vector<int32_t> deg(m);
  for (VertexId v = 0; v < m; v++) deg[v] = g.get_degree(v);  // out-degree, csr_graph.h:295
  gdn_stats st;
  const int ngpus = gdn_num_gpus();
  if (ngpus > 1)
    gdn_must(gdn_pr_multi(m, g.E(), g.in_rowptr(), g.in_colidx(), deg.data(), scores, kDamp, EPSILON, MAX_ITER, ngpus,
                          nullptr, &st), "PRSolver");
  else
    gdn_must(gdn_pr(m, g.E(), g.in_rowptr(), g.in_colidx(), deg.data(), scores, kDamp, EPSILON, MAX_ITER, &st), "PRSolver");
  // the per-iteration line of src/pr/omp_base.cc:35 (golden: test/reference/graph-pr.mtx.out:13-27)
  std::vector<double> trace(MAX_ITER + 1);
  int32_t n = 0;
  gdn_must(gdn_pr_last_trace(MAX_ITER + 1, &n, trace.data()), "PRSolver");
  for (int32_t i = 0; i < n && i <= MAX_ITER; i++) printf(" %2d    %lf\n", i + 1, trace[i]);
  printf("\titerations = %d.\n", st.iterations);
  printf("\truntime [hip_mi355x] = %f ms.\n", st.solve_ms);
}
