import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from gardenia_amd import graphio, solvers
m = 100000
g = graphio.build_csr(m, np.arange(m - 1, dtype=np.int64), np.arange(1, m, dtype=np.int64))
d = np.full(m, 1000000000, np.int32)
t = time.time(); st = solvers.BFSSolver(solvers.Graph(csr=g), 0, d); t = time.time() - t
assert np.array_equal(d, np.arange(m)), "chain depths"
print("chain of %d: %d levels, solve %.1f ms (wall %.2f s), GDN_BFS_SMALL_NF=%s" % (m, st["iterations"], st["solve_ms"], t, os.environ.get("GDN_BFS_SMALL_NF")))
