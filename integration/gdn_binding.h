// gdn_binding.h -- shared by the hip_mi355x_<k>.cc wrappers a GARDENIA maintainer drops into src/<k>/.
// Each wrapper defines the ONE link-time symbol of its kernel directory (XxxSolver, e.g. src/bfs/bfs.h:43) and
// forwards to the C-ABI of include/gardenia_hip.h; main.cc, verifier.cc and csr_graph.h stay untouched.  Failures end
// the process the way the reference's CUDA solvers do (CUDA_SAFE_CALL, include/cutil_subset.h:4-12).
#pragma once
#include <cstdio>
#include <cstdlib>

#include "gardenia_hip.h"

static inline void gdn_must(int status, const char *what) {
  if (status != GDN_OK) {
    fprintf(stderr, "%s: %s\n", what, gdn_last_error());
    exit(EXIT_FAILURE);
  }
}

// GDN_NUM_GPUS: the devices PRSolver / SpmvSolver spread over, read like the OpenMP solvers read OMP_NUM_THREADS
static inline int gdn_num_gpus() {
  const char *e = getenv("GDN_NUM_GPUS");
  const int n = e ? atoi(e) : 1;
  return n < 1 ? 1 : n;
}
