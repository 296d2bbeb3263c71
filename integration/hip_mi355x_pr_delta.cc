// src/pr/hip_mi355x_delta.cc -- the delta-PageRank PRSolver (chosen at link time like src/pr/delta.cu:140 /
// src/pr/omp_delta.cc:52) on MI355X.
#include "pr.h"
#include "gdn_binding.h"

void PRSolver(Graph &g, ScoreT *scores) {
  gdn_stats st;
  gdn_must(gdn_pr_delta(g.V(), g.E(), g.in_rowptr(), g.in_colidx(), g.out_rowptr(), g.out_colidx(), scores, kDamp, EPSILON,
                        epsilon2, MAX_ITER, /*push_div (src/pr/delta.cu:178)=*/8, &st), "PRSolver");
  printf("\titerations = %d.\n", st.iterations);
  printf("\truntime [hip_mi355x_delta] = %f ms.\n", st.solve_ms);
}
