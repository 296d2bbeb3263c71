"""The worklist push primitives on their own (gdn_worklist_filter_dev -> gdn_wl_push / gdn_wl_push_staged,
gardenia_amd/csrc/gdn_common.hpp; replaces Worklist::push / Worklist2::push_1item, include/worklistc.h:44-89): the queue
holds exactly the flagged indices (any order), the count is exact, and a queue that is too small flags the overflow
instead of dropping silently."""
import ctypes as C

import numpy as np
import pytest

from gardenia_amd import _cabi

pytestmark = pytest.mark.gpu


def run(flags, capacity, staged):
    L = _cabi.lib()
    n = len(flags)
    bufs = [C.c_void_p() for _ in range(4)]
    sizes = [max(4 * n, 4), max(4 * capacity, 4), 4, 4]
    for b, sz in zip(bufs, sizes):
        _cabi.check(L.gdn_dev_alloc(sz, C.byref(b)))
    d_flags, d_queue, d_count, d_over = bufs
    try:
        if n:
            _cabi.check(L.gdn_dev_upload(d_flags, flags.ctypes.data_as(C.c_void_p), 4 * n))
        guard = np.full(max(capacity, 1), -7, np.int32)
        _cabi.check(L.gdn_dev_upload(d_queue, guard.ctypes.data_as(C.c_void_p), 4 * max(capacity, 1)))
        _cabi.check(L.gdn_worklist_filter_dev(d_flags, n, staged, d_queue, capacity, d_count, d_over))
        cnt, over = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
        _cabi.check(L.gdn_dev_download(cnt.ctypes.data_as(C.c_void_p), d_count, 4))
        _cabi.check(L.gdn_dev_download(over.ctypes.data_as(C.c_void_p), d_over, 4))
        q = np.empty(max(capacity, 1), np.int32)
        _cabi.check(L.gdn_dev_download(q.ctypes.data_as(C.c_void_p), d_queue, 4 * max(capacity, 1)))
        return int(cnt[0]), int(over[0]), q[:capacity]
    finally:
        for b in bufs:
            L.gdn_dev_free(b)


@pytest.mark.parametrize("staged", [0, 1])
@pytest.mark.parametrize("n,density", [(0, 0.5), (1, 1.0), (63, 0.5), (64, 1.0), (65, 0.0), (1000, 0.01), (100003, 0.3),
                                       (1 << 20, 0.9), ((1 << 20) + 17, 1.0)])
def test_filter_pushes_exactly_the_flagged_indices(n, density, staged):
    rng = np.random.default_rng(n + 3)
    flags = (rng.random(n) < density).astype(np.int32) * rng.integers(1, 100, n).astype(np.int32)
    want = np.nonzero(flags)[0].astype(np.int32)
    cnt, over, q = run(flags, n, staged)
    assert cnt == len(want) and over == 0
    assert np.array_equal(np.sort(q[:cnt]), want)
    assert (q[cnt:] == -7).all()  # nothing written behind the count


@pytest.mark.parametrize("staged", [0, 1])
def test_filter_flags_overflow(staged):
    n = 50000
    flags = np.ones(n, np.int32)
    cap = 12345
    cnt, over, q = run(flags, cap, staged)
    assert over == 1 and cnt == n  # the count says how many were pushed; only `capacity` of them were stored
    assert len(np.unique(q)) == cap and q.min() >= 0 and q.max() < n
