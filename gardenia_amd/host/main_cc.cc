// cc main: same CLI and flow as src/cc/main.cc:5-19 (argv[3]/argv[4] are optional here; the
// reference reads them unconditionally)
#include <cstdlib>
#include <iostream>

#include "gardenia_host.hpp"

static int real_main(int argc, char *argv[]) {
  std::cout << "Connected Component (gardenia_amd, MI355X)\n";
  if (argc < 3) {
    printf("Usage: %s <filetype> <graph> [symmetrize(0/1)] [reverse(0/1)]\n", argv[0]);
    return 1;
  }
  Graph g(argv[2], argv[1], argc > 3 ? atoi(argv[3]) : 0, argc > 4 ? atoi(argv[4]) : 0);
  std::vector<CompT> comp(g.V());
  for (int i = 0; i < g.V(); i++) comp[i] = i;
  CCSolver(g, &comp[0]);
  return CCVerifier(g, &comp[0]) ? 0 : 2;
}

int main(int argc, char *argv[]) { return gardenia_guarded_main(real_main, argc, argv); }
