"""gardenia_amd -- MI355X-native drop-in for the CSR hot path of the GARDENIA benchmark.

Layout: csrc/ (HIP kernels + the C-ABI of include/gardenia_hip.h), host/ (C++ mirror of the
reference's Graph / XxxSolver / main.cc harness), solvers.py + graphio.py (numpy-side mirror
used by tests and bench.py).  No CPU compute path lives in this package.
"""
__all__ = ["graphio", "solvers"]
