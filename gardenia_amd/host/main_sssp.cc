// sssp main: same CLI and flow as src/sssp/main.cc:5-30 (all weights 1, delta default 1)
#include <cstdlib>
#include <iostream>

#include "gardenia_host.hpp"

static int real_main(int argc, char *argv[]) {
  std::cout << "Single Source Shortest Path (gardenia_amd, MI355X)\n";
  if (argc < 3) {
    std::cout << "Usage: " << argv[0] << " <filetype> <graph-prefix> [symmetrize(0/1)] [reverse(0/1)] [source_id(0)] [delta(1)]\n";
    return 1;
  }
  int delta = 1, source = 0;
  bool symmetrize = false, need_reverse = false;
  if (argc > 3) symmetrize = atoi(argv[3]);
  if (argc > 4) need_reverse = atoi(argv[4]);
  Graph g(argv[2], argv[1], symmetrize, need_reverse);
  if (argc > 5) source = atoi(argv[5]);
  if (argc > 6) delta = atoi(argv[6]);
  std::vector<DistT> distances(g.V(), (DistT)kDistInf);
  std::vector<DistT> wt(g.E(), DistT(1));
  SSSPSolver(g, source, &wt[0], &distances[0], delta);
  return SSSPVerifier(g, source, &wt[0], &distances[0]) ? 0 : 2;
}

int main(int argc, char *argv[]) { return gardenia_guarded_main(real_main, argc, argv); }
