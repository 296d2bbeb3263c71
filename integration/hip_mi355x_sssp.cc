// src/sssp/hip_mi355x.cc -- SSSPSolver (src/sssp/sssp.h:47, called by src/sssp/main.cc:27) on MI355X.
#include "sssp.h"
#include "gdn_binding.h"

void SSSPSolver(Graph &g, int source, DistT *weight, DistT *dist, int delta) {
  gdn_stats st;
  gdn_must(gdn_sssp(g.V(), g.E(), g.out_rowptr(), g.out_colidx(), weight, source, delta, dist, &st), "SSSPSolver");
  printf("\titerations = %d.\n", st.iterations);
  printf("\truntime [hip_mi355x] = %f ms.\n", st.solve_ms);
}
