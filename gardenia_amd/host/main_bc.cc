// bc main: same CLI and flow as src/bc/main.cc:5-25
#include <cstdlib>
#include <iostream>

#include "gardenia_host.hpp"

static int real_main(int argc, char *argv[]) {
  std::cout << "Betweenness Centrality (gardenia_amd, MI355X)\n";
  if (argc < 3) {
    std::cout << "Usage: " << argv[0] << " <filetype> <graph-prefix> [symmetrize(0/1)] [reverse(0/1)] [source_id(0)]\n";
    std::cout << "Example: " << argv[0] << " mtx web-Google\n";
    return 1;
  }
  bool symmetrize = false, need_reverse = false;
  if (argc > 3) symmetrize = atoi(argv[3]);
  if (argc > 4) need_reverse = atoi(argv[4]);
  Graph g(argv[2], argv[1], symmetrize, need_reverse);
  int source = 0;
  if (argc == 6) source = atoi(argv[5]);
  std::vector<ScoreT> scores(g.V(), 0);
  BCSolver(g, source, &scores[0]);
  return BCVerifier(g, source, 1, &scores[0]) ? 0 : 2;
}

int main(int argc, char *argv[]) { return gardenia_guarded_main(real_main, argc, argv); }
