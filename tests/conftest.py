import os
import sys

# The suite forces big-graph code paths onto small graphs and compares layout variants bit for bit through TEST HOOKS of the
# library (thresholds such as GDN_BFS_HEADS_MIN_NNZ, GDN_PB_HUB_MIN_NNZ ...).  A production process does not read them: they are
# honoured only while GDN_TEST_HOOKS=1 (csrc/gdn_common.hpp, gdn_test_option); child processes of the tests inherit it.
os.environ.setdefault("GDN_TEST_HOOKS", "1")

import numpy as np
import pytest

try:  # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's): load it FIRST so that
    import torch  # noqa: F401  libgardenia_hip.so binds to the same HIP runtime instance in this process
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def csr_from(d, prefix=""):
    from gardenia_amd.graphio import CSR
    rp = np.ascontiguousarray(d[prefix + "rowptr"], dtype=np.uint64)
    ci = np.ascontiguousarray(d[prefix + "colidx"], dtype=np.int32)
    return CSR(len(rp) - 1, rp, ci)


@pytest.fixture(scope="session")
def orc():
    from oracle import binding
    binding.lib()
    return binding
