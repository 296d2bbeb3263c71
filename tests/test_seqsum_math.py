"""The arithmetic of gardenia_amd/csrc/gdn_seqsum.hpp (GDN_PR_SUM=reference): the reference's row sum -- src/pr/omp_base.cc:27-30,
fl(S + x_k) one addition after the other -- evaluated by parallel scans of parity functions, a block of 64 x N elements at a time.
This is a line-for-line CPU emulation of seq_quant / seq_push / seq_compose / seq_block (integers only, no float arithmetic but the
hardware-path additions), checked bit for bit against numpy's sequential float32 sum on inputs chosen to hit every branch: equal
terms, exact ties, powers of two, zeros, denormals, a term larger than the running sum, negative terms, inf and nan.  The GPU kernel
itself is tested by tests/test_gpu_parity.py::test_pr_reference_order_sums_*."""
import numpy as np
import pytest

CLAMP = 1 << 23


def f2u(x):
    return int(np.float32(x).view(np.uint32))


def u2f(u):
    return np.uint32(u).view(np.float32)


def seq_quant(xb, E):
    """gdn_seqsum.hpp's seq_quant: float arithmetic that is exact -- x scaled by a power of two, its floor, the remainder"""
    with np.errstate(all="ignore"):
        y = np.ldexp(u2f(xb), 150 - E).astype(np.float32)
        fl = np.floor(y)
        fr = np.float32(y - fl)
    inside = bool(y >= 0) and bool(y < np.float32(8388608.0))
    if not inside:
        return CLAMP, 0
    return int(fl) + (1 if fr > np.float32(0.5) else 0), 1 if fr == np.float32(0.5) else 0


def seq_push(p, q, tie):
    t0, t1 = p[0] + q, p[1] + q
    return (t0 + (tie & t0), t1 + (tie & (t1 + 1)))


def seq_compose(f, g):
    return (f[0] + (g[1] if f[0] & 1 else g[0]), f[1] + (g[1] if (f[1] + 1) & 1 else g[0]))


def seq_block(S, xs, n_per_lane):
    start = 0
    while True:
        E = S >> 23
        if 1 <= E <= 254:
            ps = []
            for lane in range(64):
                p = (0, 0)
                if lane >= start:
                    for k in range(n_per_lane):
                        p = seq_push(p, *seq_quant(xs[n_per_lane * lane + k], E))
                ps.append(p)
            inc, acc = [], (0, 0)
            for lane in range(64):
                acc = seq_compose(acc, ps[lane])
                inc.append(acc)
            P0 = (S & 0x7FFFFF) | 0x800000
            tot = [P0 + (inc[lane][1] if P0 & 1 else inc[lane][0]) for lane in range(64)]
            assert max(tot) < 1 << 32  # (what SEQ_CLAMP is for)
            L = next((lane for lane in range(start, 64) if tot[lane] >= 1 << 24), None)
            if L is None:
                return (E << 23) | (tot[63] & 0x7FFFFF)
            Pb = tot[L - 1] if L else P0
            S = (E << 23) | (Pb & 0x7FFFFF)
        else:
            L = start
        with np.errstate(all="ignore"):
            for k in range(n_per_lane):
                S = f2u(u2f(S) + u2f(xs[n_per_lane * L + k]))
        start = L + 1
        if start >= 64:
            return S


def seqsum(vals, n_per_lane=8):
    blk = 64 * n_per_lane
    xs = np.zeros((len(vals) + blk - 1) // blk * blk, np.uint32)
    xs[:len(vals)] = np.asarray(vals, np.float32).view(np.uint32)
    S = 0
    for b in range(0, len(xs), blk):
        S = seq_block(S, [int(v) for v in xs[b:b + blk]], n_per_lane)
    return S


def reference(vals):
    s = np.float32(0)
    with np.errstate(all="ignore"):
        for v in np.asarray(vals, np.float32):
            s = np.float32(s + v)
    return f2u(s)


def _case(kind, n, rng):
    if kind == 0:
        return rng.random(n).astype(np.float32)
    if kind == 1:  # the first PageRank iteration: long runs of equal terms
        return np.full(n, np.float32(1.0 / 134217728 / 3), np.float32)
    if kind == 2:
        return (rng.random(n) * 1e-9).astype(np.float32)
    if kind == 3:  # magnitudes over many binades
        return np.exp(rng.normal(-20, 6, n)).astype(np.float32)
    if kind == 4:  # small integers times powers of two: exact ties
        return np.ldexp(rng.integers(1, 4, n).astype(np.float32), rng.integers(-40, -20, n)).astype(np.float32)
    if kind == 5:
        return np.where(rng.random(n) < 0.3, 0, rng.random(n) * 1e-6).astype(np.float32)
    if kind == 6:  # zeros and denormals in front
        return np.concatenate([np.zeros(5, np.float32), np.full(3, 1e-40, np.float32), (rng.random(n) * 1e-30).astype(np.float32)])
    if kind == 7:  # powers of two only
        return np.ldexp(np.float32(1.0), rng.integers(-30, -24, n)).astype(np.float32)
    if kind == 8:  # one term far above the running sum
        v = (rng.random(n) * 1e-7).astype(np.float32)
        v[rng.integers(0, n)] = np.float32(0.5)
        return v
    v = (rng.random(n) * 1e-3).astype(np.float32)  # a negative term: the hardware path
    v[n // 2] = -v[n // 2]
    return v


@pytest.mark.parametrize("n_per_lane", [4, 8])
def test_scan_of_parity_functions_equals_the_sequential_fp32_sum(n_per_lane):
    rng = np.random.default_rng(6)
    for t in range(30):
        v = _case(t % 10, int(rng.integers(1, 2500)), rng)
        assert seqsum(v, n_per_lane) == reference(v), (t, len(v))


def test_inf_and_nan_take_the_hardware_path():
    rng = np.random.default_rng(7)
    v = (rng.random(700) * 1e-3).astype(np.float32)
    v[100] = np.inf
    assert seqsum(v) == reference(v)
    v[300] = -np.inf
    got, want = seqsum(v), reference(v)
    assert np.isnan(u2f(got)) and np.isnan(u2f(want))
