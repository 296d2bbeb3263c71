// src/cc/hip_mi355x.cc -- CCSolver (src/cc/cc.h:28, called by src/cc/main.cc:16) on MI355X.
#include "cc.h"
#include "gdn_binding.h"

void CCSolver(Graph &g, CompT *comp) {
  gdn_stats st;
  // weak connectivity of a directed input wants the reverse graph too (src/cc/omp_base.cc:33-41 walks in_neigh)
  const bool rev = g.has_reverse_graph();
  gdn_must(gdn_cc(g.V(), g.E(), g.out_rowptr(), g.out_colidx(), rev ? g.in_rowptr() : nullptr,
                  rev ? g.in_colidx() : nullptr, comp, &st), "CCSolver");
  printf("\titerations = %d.\n", st.iterations);
  printf("\truntime [hip_mi355x] = %f ms.\n", st.solve_ms);
}
