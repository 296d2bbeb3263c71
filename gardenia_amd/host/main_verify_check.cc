// verify_check: runs ONE harness verifier (verifiers.cc) on a result vector read from a raw binary file -- no solver, no
// GPU.  The tests feed it correct vectors (from the CPU oracle) and corrupted ones: a verifier that printed "Correct"
// whatever it was given would make every `grep Correct` check of the mains worthless.
//   verify_check <bfs|pr|spmv|sssp|cc|tc|bc> <filetype> <graph-prefix> <symmetrize> <result.bin> [source] [aux.bin]
// result.bin: m x int32 (bfs, sssp, cc), m x float (pr, spmv, bc), 1 x uint64 (tc).  aux.bin: sssp = nnz x int32 weights;
// spmv = nnz floats Ax, then m floats x, then m floats y0.  Exit code 0 = Correct, 2 = the verifier rejected the vector.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "gardenia_host.hpp"

template <typename T>
static std::vector<T> slurp(const char *path, size_t count) {
  std::vector<T> v(count);
  std::ifstream f(path, std::ios::binary);
  if (!f || !f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(count * sizeof(T)))) {
    std::cerr << "verify_check: cannot read " << count << " items from " << path << "\n";
    exit(1);
  }
  return v;
}

static int real_main(int argc, char *argv[]) {
  if (argc < 6) {
    std::cout << "Usage: " << argv[0] << " <kernel> <filetype> <graph-prefix> <symmetrize> <result.bin> [source] [aux.bin]\n";
    return 1;
  }
  const std::string k = argv[1];
  const bool symmetrize = atoi(argv[4]) != 0;
  Graph g(argv[3], argv[2], symmetrize, /*need_reverse=*/k == "pr" || k == "spmv");
  const int source = argc > 6 ? atoi(argv[6]) : 0;
  const size_t m = (size_t)g.V(), nnz = g.E();
  bool ok = false;
  if (k == "bfs") {
    std::vector<DistT> d = slurp<DistT>(argv[5], m);
    ok = BFSVerifier(g, source, d.data());
  } else if (k == "sssp") {
    std::vector<DistT> d = slurp<DistT>(argv[5], m), w = slurp<DistT>(argv[7], nnz);
    ok = SSSPVerifier(g, source, w.data(), d.data());
  } else if (k == "cc") {
    std::vector<CompT> c = slurp<CompT>(argv[5], m);
    ok = CCVerifier(g, c.data());
  } else if (k == "pr") {
    std::vector<ScoreT> s = slurp<ScoreT>(argv[5], m);
    ok = PRVerifier(g, s.data(), EPSILON);
  } else if (k == "bc") {
    std::vector<ScoreT> s = slurp<ScoreT>(argv[5], m);
    ok = BCVerifier(g, source, 1, s.data());
  } else if (k == "spmv") {
    std::vector<ValueT> y = slurp<ValueT>(argv[5], m), aux = slurp<ValueT>(argv[7], nnz + 2 * m);
    ok = SpmvVerifier(g, aux.data(), aux.data() + nnz, aux.data() + nnz + m, y.data());
  } else if (k == "tc") {
    std::vector<uint64_t> t = slurp<uint64_t>(argv[5], 1);
    ok = TCVerifier(g, t[0]);
  } else {
    std::cout << "unknown kernel " << k << "\n";
    return 1;
  }
  return ok ? 0 : 2;
}

int main(int argc, char *argv[]) { return gardenia_guarded_main(real_main, argc, argv); }
