// ref_driver.cc -- TEST INFRASTRUCTURE ONLY (oracle side).
//
// Dump driver that is compiled TOGETHER WITH the reference's own solver / verifier
// translation units, in place under /root/reference (see oracle/Makefile; nothing of the
// reference is copied into this repository).  It plays the role of src/<kernel>/main.cc:
// build the reference's Graph, pre-initialise the label array exactly like that main,
// call the reference XxxSolver and/or XxxVerifier, and dump arrays as raw little-endian
// binaries so that tests can (a) pin the CPU restatement in oracle/gardenia_oracle.cc and
// (b) generate the golden vectors committed under tests/golden/.
//
// Usage:
//   ref_<k> solve  <filetype> <prefix> <symmetrize> <reverse> <out_prefix> [extra...]
//   ref_<k> verify <filetype> <prefix> <symmetrize> <reverse> <labels.bin>  [extra...]
// extra: bfs/sssp: source ; sssp: weights.bin (int32 x nnz, "-" = all ones) ;
//        spmv: Ax.bin x.bin y0.bin ("-" = the constants of src/spmv/main.cc:27-37)
// ref_tc <prefix> <out_prefix>            (bin graph, USE_DAG orientation applied)
//
// Outputs (solve): <out>.rowptr(u64) .colidx(i32) [.in_rowptr .in_colidx] + labels file.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#if defined(K_BFS)
#include "bfs.h"
#elif defined(K_PR) || defined(K_PRDELTA)
#include "pr.h"
#if defined(K_PRDELTA)
// the legacy raw-array entry point src/pr/omp_delta.cc:52 still defines (an overload next to pr.h:31)
void PRSolver(int m, int nnz, IndexT *row_offsets, IndexT *column_indices, IndexT *out_row_offsets,
              IndexT *out_column_indices, int *degrees, ScoreT *scores);
#endif
#elif defined(K_SPMV)
#include "spmv.h"
#elif defined(K_SSSP)
#include "sssp.h"
#elif defined(K_CC)
#include "cc.h"
#elif defined(K_TC)
#include "tc.h"
#elif defined(K_BC)
#include "bc.h"
#else
#error "define one of K_BFS K_PR K_PRDELTA K_SPMV K_SSSP K_CC K_TC K_BC"
#endif

template <typename T>
static void dump(const std::string &name, const T *p, size_t n) {
  FILE *f = fopen(name.c_str(), "wb");
  if (!f) { perror(name.c_str()); exit(2); }
  if (n) fwrite(p, sizeof(T), n, f);
  fclose(f);
}

template <typename T>
static std::vector<T> slurp(const std::string &name, size_t n) {
  std::vector<T> v(n);
  FILE *f = fopen(name.c_str(), "rb");
  if (!f) { perror(name.c_str()); exit(2); }
  size_t got = fread(v.data(), sizeof(T), n, f);
  fclose(f);
  if (got != n) { fprintf(stderr, "%s: short read %zu/%zu\n", name.c_str(), got, n); exit(2); }
  return v;
}

#if defined(K_TC)
int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <prefix> <out_prefix>\n", argv[0]); return 1; }
  Graph g(argv[1], USE_DAG);  // src/tc/main.cc:12
  std::string out = argv[2];
  dump(out + ".rowptr", g.out_rowptr(), (size_t)g.V() + 1);
  dump(out + ".colidx", g.out_colidx(), (size_t)g.E());
  uint64_t total = 0;
  TCSolver(g, total);  // src/tc/omp_base.cc:6
  dump(out + ".total", &total, 1);
  TCVerifier(g, total);  // src/tc/verifier.cc:8 (the reference main leaves it commented out)
  printf("total_num_triangles = %lu\n", (unsigned long)total);
  return 0;
}
#else
int main(int argc, char **argv) {
  if (argc < 7) { fprintf(stderr, "usage: see header of ref_driver.cc\n"); return 1; }
  std::string mode = argv[1];
  bool symmetrize = atoi(argv[4]) != 0;
  bool need_reverse = atoi(argv[5]) != 0;
  Graph g(argv[3], argv[2], symmetrize, need_reverse);
  std::string io = argv[6];
  auto m = g.V();
  auto nnz = g.E();
  (void)nnz;
  if (mode == "solve") {
    dump(io + ".rowptr", g.out_rowptr(), (size_t)m + 1);
    dump(io + ".colidx", g.out_colidx(), (size_t)nnz);
    if (g.has_reverse_graph()) {
      dump(io + ".in_rowptr", g.in_rowptr(), (size_t)m + 1);
      dump(io + ".in_colidx", g.in_colidx(), (size_t)nnz);
    }
  }
#if defined(K_BFS)
  int source = argc > 7 ? atoi(argv[7]) : 0;
  if (mode == "solve") {
    std::vector<DistT> distances(m, MYINFINITY);  // src/bfs/main.cc:21
    BFSSolver(g, source, &distances[0]);
    dump(io + ".dist", distances.data(), (size_t)m);
    BFSVerifier(g, source, &distances[0]);
  } else {
    auto d = slurp<DistT>(io, (size_t)m);
    BFSVerifier(g, source, d.data());
  }
#elif defined(K_PR)
  if (mode == "solve") {
    const ScoreT init_score = 1.0f / m;  // src/pr/main.cc:17
    std::vector<ScoreT> scores(m, init_score);
    PRSolver(g, &scores[0]);
    dump(io + ".scores", scores.data(), (size_t)m);
    PRVerifier(g, &scores[0], EPSILON);
  } else {
    auto s = slurp<ScoreT>(io, (size_t)m);
    PRVerifier(g, s.data(), EPSILON);
  }
#elif defined(K_PRDELTA)
  {  // needs <reverse> = 1; offsets narrowed to the IndexT (int) arrays of the legacy signature
    std::vector<IndexT> irp(m + 1), ici(nnz), orp(m + 1), oci(nnz);
    std::vector<int> degrees(m);
    for (size_t i = 0; i <= (size_t)m; i++) { irp[i] = (IndexT)g.in_rowptr()[i]; orp[i] = (IndexT)g.out_rowptr()[i]; }
    for (size_t i = 0; i < (size_t)nnz; i++) { ici[i] = g.in_colidx()[i]; oci[i] = g.out_colidx()[i]; }
    for (size_t i = 0; i < (size_t)m; i++) degrees[i] = orp[i + 1] - orp[i];
    const ScoreT init_score = 1.0f / m;  // src/pr/main.cc:17
    std::vector<ScoreT> scores(m, init_score);
    PRSolver((int)m, (int)nnz, irp.data(), ici.data(), orp.data(), oci.data(), degrees.data(), scores.data());
    dump(io + ".scores", scores.data(), (size_t)m);
  }
#elif defined(K_SPMV)
  std::vector<ValueT> Ax(nnz, 0.2), x(m, 0.3), y0(m, 0.0);  // src/spmv/main.cc:29-36
  if (argc > 9 && std::string(argv[7]) != "-") {
    Ax = slurp<ValueT>(argv[7], nnz);
    x = slurp<ValueT>(argv[8], m);
    y0 = slurp<ValueT>(argv[9], m);
  }
  if (mode == "solve") {
    std::vector<ValueT> y(y0);
    SpmvSolver(g, Ax.data(), x.data(), y.data());
    dump(io + ".y", y.data(), (size_t)m);
    SpmvVerifier(g, Ax.data(), x.data(), y0.data(), y.data());
  } else {
    auto y = slurp<ValueT>(io, (size_t)m);
    SpmvVerifier(g, Ax.data(), x.data(), y0.data(), y.data());
  }
#elif defined(K_SSSP)
  // Only the verifier (serial Dijkstra) of the reference builds here: src/sssp/omp_base.cc
  // includes sim.h -> gem5/m5ops.h, which this image lacks.
  int source = argc > 7 ? atoi(argv[7]) : 0;
  std::vector<DistT> wt(nnz, DistT(1));  // src/sssp/main.cc:26
  if (argc > 8 && std::string(argv[8]) != "-") wt = slurp<DistT>(argv[8], nnz);
  auto d = slurp<DistT>(io, (size_t)m);
  SSSPVerifier(g, source, wt.data(), d.data());
#elif defined(K_BC)
  int source = argc > 7 ? atoi(argv[7]) : 0;
  if (mode == "solve") {
    std::vector<ScoreT> scores(m, 0);  // src/bc/main.cc:21
    BCSolver(g, source, &scores[0]);
    dump(io + ".scores", scores.data(), (size_t)m);
    BCVerifier(g, source, 1, &scores[0]);  // src/bc/main.cc:23
  } else {
    auto s = slurp<ScoreT>(io, (size_t)m);
    BCVerifier(g, source, 1, s.data());
  }
#elif defined(K_CC)
  if (mode == "solve") {
    std::vector<CompT> comp(m);
    for (int i = 0; i < m; i++) comp[i] = i;  // src/cc/main.cc:15
    CCSolver(g, &comp[0]);
    dump(io + ".comp", comp.data(), (size_t)m);
    CCVerifier(g, &comp[0]);
  } else {
    auto c = slurp<CompT>(io, (size_t)m);
    CCVerifier(g, c.data());
  }
#endif
  return 0;
}
#endif
