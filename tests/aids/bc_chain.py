import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gardenia_amd import graphio, solvers
m = 50000
g = graphio.build_csr(m, np.arange(m - 1, dtype=np.int64), np.arange(1, m, dtype=np.int64))
sc = np.zeros(m, np.float32)
st = solvers.BCSolver(solvers.Graph(csr=g), 0, sc)
print("BC on a chain of %d: %d levels, solve %.1f ms, GDN_BC_SMALL_NF=%s, max score %.3f" % (m, st["iterations"], st["solve_ms"], os.environ.get("GDN_BC_SMALL_NF"), float(np.nanmax(sc))))
