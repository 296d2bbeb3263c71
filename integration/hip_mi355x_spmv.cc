// src/spmv/hip_mi355x.cc -- SpmvSolver (src/spmv/spmv.h:29, called by src/spmv/main.cc:39) on MI355X.
#include "spmv.h"
#include "gdn_binding.h"

void SpmvSolver(Graph &g, const ValueT *Ax, const ValueT *x, ValueT *y) {
  gdn_stats st;
  const int ngpus = gdn_num_gpus();
  // the CSR main.cc weights with in_weights (src/spmv/main.cc:25,39) and verifier.cc walks: the in-CSR when the graph
  // was loaded with a reverse graph, else the graph itself (csr_graph.h:305-306 alias it)
  if (ngpus > 1)
    gdn_must(gdn_spmv_multi(g.V(), g.E(), g.in_rowptr(), g.in_colidx(), Ax, x, y, ngpus, nullptr, &st), "SpmvSolver");
  else
    gdn_must(gdn_spmv(g.V(), g.E(), g.in_rowptr(), g.in_colidx(), Ax, x, y, &st), "SpmvSolver");
  printf("\truntime [hip_mi355x] = %f ms.\n", st.solve_ms);
}
