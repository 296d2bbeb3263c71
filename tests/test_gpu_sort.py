"""gdn_sort_u64_dev (gardenia_amd/csrc/gdn_sort.hip): the radix sort under every graph / layout build, against numpy's
stable sort on the same bit field -- sizes around the tile (4096) and span (65536) boundaries, partial bit ranges
(stability is what pb_build's chunk-only sort relies on), skewed digits."""
import ctypes as C

import numpy as np
import pytest

from gardenia_amd import _cabi

pytestmark = pytest.mark.gpu


def dev_sort(keys, b0, b1):
    L = _cabi.lib()
    n = len(keys)
    a, b = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_dev_alloc(max(8 * n, 8), C.byref(a)))
    _cabi.check(L.gdn_dev_alloc(max(8 * n, 8), C.byref(b)))
    try:
        if n:
            _cabi.check(L.gdn_dev_upload(a, keys.ctypes.data_as(C.c_void_p), 8 * n))
        out = C.c_void_p()
        _cabi.check(L.gdn_sort_u64_dev(a, b, n, b0, b1, C.byref(out)))
        assert out.value in (a.value, b.value)
        res = np.empty(n, np.uint64)
        if n:
            _cabi.check(L.gdn_dev_download(res.ctypes.data_as(C.c_void_p), out, 8 * n))
        return res
    finally:
        L.gdn_dev_free(a)
        L.gdn_dev_free(b)


def want(keys, b0, b1):
    if b1 <= b0:
        return keys
    field = (keys >> np.uint64(b0)) & np.uint64((1 << (b1 - b0)) - 1 if b1 - b0 < 64 else 0xFFFFFFFFFFFFFFFF)
    return keys[np.argsort(field, kind="stable")]


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 4095, 4096, 4097, 65535, 65536, 65537, 200001, (1 << 20) + 7])
def test_sort_sizes_full_range(n):
    rng = np.random.default_rng(n + 1)
    keys = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    got = dev_sort(keys, 0, 64)
    assert np.array_equal(got, np.sort(keys))


@pytest.mark.parametrize("b0,b1", [(0, 1), (0, 8), (5, 17), (20, 32), (40, 59), (47, 59), (3, 3), (56, 64)])
def test_sort_bit_ranges_are_stable(b0, b1):
    rng = np.random.default_rng(b0 * 64 + b1)
    keys = rng.integers(0, 1 << 62, 300007, dtype=np.uint64)
    got = dev_sort(keys, b0, b1)
    assert np.array_equal(got, want(keys, b0, b1))


@pytest.mark.parametrize("kind", ["one_digit", "two_values", "sorted", "reversed", "rmat_like"])
def test_sort_skewed_digits(kind):
    rng = np.random.default_rng(7)
    n = 500003
    if kind == "one_digit":
        keys = (np.uint64(0xAB) << np.uint64(24)) | rng.integers(0, 1 << 24, n, dtype=np.uint64)
    elif kind == "two_values":
        keys = rng.integers(0, 2, n, dtype=np.uint64) * np.uint64(0xFFFF0000FFFF)
    elif kind == "sorted":
        keys = np.sort(rng.integers(0, 1 << 40, n, dtype=np.uint64))
    elif kind == "reversed":
        keys = np.sort(rng.integers(0, 1 << 40, n, dtype=np.uint64))[::-1].copy()
    else:  # heavy low ids in both halves, like (row << 32 | col) of an R-MAT edge list
        r = (rng.random(n) ** 6 * (1 << 20)).astype(np.uint64)
        c = (rng.random(n) ** 6 * (1 << 20)).astype(np.uint64)
        keys = (r << np.uint64(32)) | c
    assert np.array_equal(dev_sort(keys, 0, 52), np.sort(keys))
    assert np.array_equal(dev_sort(keys, 32, 52), want(keys, 32, 52))
