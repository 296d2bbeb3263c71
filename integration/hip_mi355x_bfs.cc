// src/bfs/hip_mi355x.cc -- BFSSolver (src/bfs/bfs.h:43, called by src/bfs/main.cc:22) on MI355X.
#include "bfs.h"
#include "gdn_binding.h"

void BFSSolver(Graph &g, int source, DistT *dist) {
  gdn_stats st;
  const bool rev = g.has_reverse_graph();  // csr_graph.h:302; the bottom-up levels want the in-CSR
  gdn_must(gdn_bfs(g.V(), g.E(), g.out_rowptr(), g.out_colidx(), rev ? g.in_rowptr() : nullptr,
                   rev ? g.in_colidx() : nullptr, source, dist, &st), "BFSSolver");
  printf("\titerations = %d.\n", st.iterations);  // the lines of src/bfs/omp_base.cc:58-59
  printf("\truntime [hip_mi355x] = %f ms.\n", st.solve_ms);
}
