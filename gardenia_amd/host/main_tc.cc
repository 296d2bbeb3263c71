// tc main: same flow as src/tc/main.cc:5-22 (bin graph prefix; USE_DAG orientation is applied on
// the device inside TCSolver); optional "mtx <prefix>" form symmetrizes a Matrix Market file.
#include <cstdlib>
#include <iostream>
#include <string>

#include "gardenia_host.hpp"

static int real_main(int argc, char *argv[]) {
  if (argc < 2) {
    printf("Usage: %s <graph-prefix>   |   %s mtx <graph-prefix>\n", argv[0], argv[0]);
    return 1;
  }
  std::cout << "Triangle Counting (gardenia_amd, MI355X)\nUsing DAG (static orientation)\n";
  const bool mtx = argc > 2 && std::string(argv[1]) == "mtx";
  Graph g(mtx ? argv[2] : argv[1], mtx ? "mtx" : "bin", mtx, false);
  std::cout << "|V| " << g.size() << " |E| " << g.sizeEdges() << "\n";
  uint64_t total = 0;
  TCSolver(g, total);
  std::cout << "total_num_triangles = " << total << "\n";
  return TCVerifier(g, total) ? 0 : 2;
}

int main(int argc, char *argv[]) { return gardenia_guarded_main(real_main, argc, argv); }
