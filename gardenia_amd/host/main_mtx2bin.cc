// mtx2bin -- .mtx -> the binary CSR triple <prefix>.meta.txt / .vertex.bin / .edge.bin that
// `Graph(prefix, "bin")` reads (include/csr_graph.h:219-230, src/common/graph.cc:7-18).  The reference ships
// readers of this format only (tools/converter.cc:778-792 converts other formats); every kernel main takes it.
// Clean-up = the mtx loader's (csr_graph.h:74-169): self loops and duplicates dropped, rows ascending,
// optional symmetrization; the CSR is built on the device (gdn_graph_from_edges).
#include <cstdlib>

#include "gardenia_host.hpp"

static int run(int argc, char **argv) {
  if (argc < 3) {
    std::cout << "Usage: " << argv[0] << " <mtx prefix (without .mtx)> <output prefix> [symmetrize(0/1)] [build on host(0/1)]\n";
    return 1;
  }
  const bool symmetrize = argc > 3 && atoi(argv[3]) != 0;
  const bool on_host = argc > 4 && atoi(argv[4]) != 0;
  Graph g(argv[1], "mtx", symmetrize, false, /*device_ingest=*/!on_host);
  g.write_bin(argv[2]);
  std::cout << "wrote " << argv[2] << ".meta.txt/.vertex.bin/.edge.bin: |V| " << g.V() << " |E| " << g.E() << " max_degree "
            << g.get_max_degree() << "\n";
  return 0;
}

int main(int argc, char **argv) { return gardenia_guarded_main(run, argc, argv); }
