// spmv main: same CLI and flow as src/spmv/main.cc:5-42 (Ax = 0.2, x = 0.3, y = 0)
#include <cstdlib>
#include <iostream>

#include "gardenia_host.hpp"

static int real_main(int argc, char *argv[]) {
  printf("Sparse Matrix-Vector Multiplication (gardenia_amd, MI355X)\n");
  if (argc < 3) {
    std::cout << "Usage: " << argv[0] << " <filetype> <graph-prefix> [symmetrize(0/1)] [reverse(0/1)]\n";
    return 1;
  }
  bool symmetrize = false, need_reverse = false;
  if (argc > 3) symmetrize = atoi(argv[3]);
  if (argc > 4) need_reverse = atoi(argv[4]);
  Graph g(argv[2], argv[1], symmetrize, need_reverse);
  std::vector<ValueT> Ax(g.E(), 0.2f), x(g.V(), 0.3f), y(g.V(), 0.0f), y0(g.V(), 0.0f);
  SpmvSolver(g, Ax.data(), x.data(), y.data());
  return SpmvVerifier(g, Ax.data(), x.data(), y0.data(), y.data()) ? 0 : 2;
}

int main(int argc, char *argv[]) { return gardenia_guarded_main(real_main, argc, argv); }
