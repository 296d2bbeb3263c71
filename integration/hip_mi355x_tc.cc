// src/tc/hip_mi355x.cc -- TCSolver (src/tc/tc.h:7, called by src/tc/main.cc:17) on MI355X.  `Graph` here is
// include/graph.hh, already oriented by `Graph g(prefix, USE_DAG)` (src/tc/main.cc:12): oriented = 1.
#include "tc.h"
#include "gdn_binding.h"

void TCSolver(Graph &g, uint64_t &total) {
  gdn_stats st;
  // eidType is int64_t (common.h:36): the same bits as the ABI's uint64_t offsets
  gdn_must(gdn_tc(g.V(), (uint64_t)g.E(), reinterpret_cast<const uint64_t *>(g.out_rowptr()), g.out_colidx(), /*oriented=*/1,
                  &total, &st), "TCSolver");
  printf("\truntime [hip_mi355x] = %f sec\n", st.solve_ms * 1e-3);  // src/tc/omp_base.cc prints seconds
}
