// pr_delta main: the CLI and flow of src/pr/main.cc:5-22 with the delta variant of the solver linked in
// (src/pr/delta.cu, `make pr_delta` in the reference).  That variant stops when its frontier is empty -- every
// |delta| <= epsilon2 * score (pr.h:8) -- which happens before the L1 change reaches EPSILON, so the verifier's
// one-iteration residual is checked against 20 * EPSILON here.
#include <cstdlib>
#include <iostream>

#include "gardenia_host.hpp"

static int real_main(int argc, char *argv[]) {
  std::cout << "Delta PageRank (gardenia_amd, MI355X)\n";
  if (argc < 3) {
    std::cout << "Usage: " << argv[0] << " <filetype> <graph-prefix> [symmetrize(0/1)]\n";
    return 1;
  }
  bool symmetrize = false;
  if (argc > 3) symmetrize = atoi(argv[3]);
  Graph g(argv[2], argv[1], symmetrize, 1);
  const ScoreT init_score = 1.0f / g.V();
  std::vector<ScoreT> scores(g.V(), init_score);
  PRDeltaSolver(g, &scores[0]);
  return PRVerifier(g, &scores[0], 20 * EPSILON) ? 0 : 2;
}

int main(int argc, char *argv[]) { return gardenia_guarded_main(real_main, argc, argv); }
