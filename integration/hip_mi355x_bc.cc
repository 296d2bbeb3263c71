// src/bc/hip_mi355x.cc -- BCSolver (src/bc/bc.h:37, called by src/bc/main.cc:22) on MI355X.
#include "bc.h"
#include "gdn_binding.h"

void BCSolver(Graph &g, int source, ScoreT *scores) {
  gdn_stats st;
  gdn_must(gdn_bc(g.V(), g.E(), g.out_rowptr(), g.out_colidx(), source, scores, &st), "BCSolver");
  printf("\titerations = %d.\n", st.iterations);
  printf("\truntime [hip_mi355x] = %f ms.\n", st.solve_ms);
}
