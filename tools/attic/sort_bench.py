"""gdn_sort_u64_dev throughput: python tools/sort_bench.py [log2 n] [end_bit]  (random keys of end_bit bits)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from gardenia_amd import _cabi  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
end = int(sys.argv[2]) if len(sys.argv) > 2 else 59
n = 1 << lg
L = _cabi.lib()
keys = np.random.default_rng(1).integers(0, 1 << end, n, dtype=np.uint64)
a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
for p in (a, b, c):
    _cabi.check(L.gdn_dev_alloc(8 * n, C.byref(p)))
_cabi.check(L.gdn_dev_upload(c, keys.ctypes.data_as(C.c_void_p), 8 * n))
for b0, b1 in ((0, end), (end - 12, end)):
    best = 1e30
    for _ in range(3):
        _cabi.check(L.gdn_dev_upload(a, keys.ctypes.data_as(C.c_void_p), 8 * n))
        out = C.c_void_p()
        t = time.perf_counter()
        _cabi.check(L.gdn_sort_u64_dev(a, b, n, b0, b1, C.byref(out)))
        best = min(best, time.perf_counter() - t)
    print("2^%d keys, bits [%d, %d): %.2f ms = %.2f G keys/s, %.1f ms per 8-bit pass" % (
        lg, b0, b1, best * 1e3, n / best / 1e9, best * 1e3 / ((b1 - b0 + 7) // 8)))
