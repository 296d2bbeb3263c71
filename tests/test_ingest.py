"""Graph ingest: the parallel .mtx parser of the C++ host mirror, the mtx -> bin converter and the
device CSR builder gdn_graph_from_edges, against the numpy restatement of the reference loader
(include/csr_graph.h:74-169: self loops dropped, rows ascending, duplicates dropped, optional symmetrization)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from gardenia_amd import _cabi, graphio

BIN = os.path.join(ROOT, "gardenia_amd", "host", "bin")
G = os.path.join(GOLDEN, "graphs")


def run(exe, *args):
    p = subprocess.run([os.path.join(BIN, exe), *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    return p.returncode, p.stdout


def same(a, b):
    return a.m == b.m and np.array_equal(a.rowptr, b.rowptr) and np.array_equal(a.colidx, b.colidx)


def write_messy_mtx(path, m, src, dst, rng):
    """1-based pairs with everything the reference parser tolerates: banner + comment lines, a third (weight)
    column, tabs, CRLF endings, blank lines, '#' lines, self loops, duplicates."""
    with open(path, "w", newline="") as f:
        f.write("%%MatrixMarket matrix coordinate pattern general\n% a comment\n")
        f.write(f"{m} {m} {len(src)}\n")
        for i, (a, b) in enumerate(zip(src.tolist(), dst.tolist())):
            k = int(rng.integers(0, 8))
            if k == 0:
                f.write(f"{a + 1}\t{b + 1}\r\n")
            elif k == 1:
                f.write(f"{a + 1} {b + 1} 0.5\n")
            elif k == 2:
                f.write(f"  {a + 1}   {b + 1}\n")
            else:
                f.write(f"{a + 1} {b + 1}\n")
            if i % 97 == 0:
                f.write("\n# not an edge\n")


@pytest.mark.parametrize("sym", [0, 1])
def test_mtx2bin_host_build_matches_the_loader_semantics(tmp_path, sym):
    """CPU: parser + host CSR build (4th argument 1 = build on the host; the device build is the GPU test below)."""
    rng = np.random.default_rng(7 + sym)
    m = 3000
    n = 40000
    src = rng.integers(0, m, n)
    dst = np.where(rng.random(n) < 0.05, src, rng.integers(0, m, n))  # 5 % self loops
    src = np.concatenate([src, src[:5000]])  # duplicates
    dst = np.concatenate([dst, dst[:5000]])
    write_messy_mtx(str(tmp_path / "g.mtx"), m, src, dst, rng)
    rc, out = run("mtx2bin", tmp_path / "g", tmp_path / "out", sym, 1)
    assert rc == 0, out
    got = graphio.read_bin(str(tmp_path / "out"))
    s, d = (np.concatenate([src, dst]), np.concatenate([dst, src])) if sym else (src, dst)
    want = graphio.build_csr(m, s, d)
    assert same(got, want)
    assert same(got, graphio.read_mtx(str(tmp_path / "g.mtx"), bool(sym)))
    meta = open(str(tmp_path / "out.meta.txt")).read().split()
    assert [int(x) for x in meta] == [m, want.nnz, 4, int(want.degrees().max())]


def test_mtx2bin_on_reference_fixtures_host(tmp_path):
    for name, sym, nnz in (("test_bc", 1, 26), ("test_cc", 1, 36), ("chesapeake", 1, 340), ("test_bc", 0, 15)):
        rc, out = run("mtx2bin", os.path.join(G, name), tmp_path / name, sym, 1)
        assert rc == 0 and f"|E| {nnz}" in out, out  # BASELINE.md known answers
        assert same(graphio.read_bin(str(tmp_path / name)), graphio.read_mtx(os.path.join(G, name + ".mtx"), bool(sym)))


def test_large_file_is_parsed_in_parallel_pieces(tmp_path):
    """> 1 MiB of text takes the multi-threaded path: pieces are cut at line boundaries."""
    rng = np.random.default_rng(3)
    m, n = 50000, 200000
    src, dst = rng.integers(0, m, n), rng.integers(0, m, n)
    with open(tmp_path / "big.mtx", "w") as f:
        f.write(f"{m} {m} {n}\n")
        f.write("".join(f"{a + 1} {b + 1}\n" for a, b in zip(src.tolist(), dst.tolist())))
    assert os.path.getsize(tmp_path / "big.mtx") > (1 << 20)
    env = dict(os.environ, OMP_NUM_THREADS="7")
    p = subprocess.run([os.path.join(BIN, "mtx2bin"), str(tmp_path / "big"), str(tmp_path / "o"), "0", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout
    assert same(graphio.read_bin(str(tmp_path / "o")), graphio.build_csr(m, src, dst))


def test_device_ingest_fails_loudly_without_gpu(tmp_path):
    if _cabi.device_count() > 0:
        pytest.skip("a GPU is present")
    rc, out = run("mtx2bin", os.path.join(G, "test_bc"), tmp_path / "x", 0)
    assert rc != 0 and "no HIP device" in out
    with pytest.raises(_cabi.GardeniaError) as ei:
        graphio.build_csr_device(4, np.array([0, 1]), np.array([1, 2]))
    assert ei.value.status == _cabi.GDN_ERR_NO_DEVICE


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("sym", [False, True])
@pytest.mark.parametrize("m,n,seed", [(1, 0, 0), (5, 3, 1), (1000, 20000, 2), (1 << 17, 1 << 21, 3)])
def test_device_builder_matches_numpy(m, n, seed, sym):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, m, n)
    dst = np.where(rng.random(n) < 0.1, src, rng.integers(0, m, n))
    if n:
        src = np.concatenate([src, src[: n // 3]])
        dst = np.concatenate([dst, dst[: n // 3]])
    got = graphio.build_csr_device(m, src, dst, sym)
    s, d = (np.concatenate([src, dst]), np.concatenate([dst, src])) if sym else (src, dst)
    assert same(got, graphio.build_csr(m, s, d))


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["front", "back", "middle", "islands"])
def test_device_builder_with_long_runs_of_empty_rows(where):
    """Rows without an edge take the offset of the next row that has one.  Short runs are filled by the thread of that row's
    first key; a run of more than 256 rows is listed and filled by a grid of its own (keys_fill_gaps_kernel) -- one thread
    storing 4 M offsets was 100 ms per CSR of the rank-ordered DAG of the forward triangle count, whose first half are the
    vertices of degree 0.  Runs in front of the first edge, behind the last one, in the middle, and many of every length."""
    m = 1 << 22
    rng = np.random.default_rng(44)
    if where == "islands":  # blocks of 1 .. 3000 vertices with edges among themselves, 1 .. 3000 empty rows between them
        src, dst, v = [], [], 0
        while v < m - 8000:
            w = int(rng.integers(1, 3000))
            e = rng.integers(v, v + w, 2 * w)
            src.append(e)
            dst.append(rng.integers(v, v + w, 2 * w))
            v += w + int(rng.integers(1, 3000))
        src, dst = np.concatenate(src), np.concatenate(dst)
    else:
        lo, hi = {"front": (m - 5000, m), "back": (0, 5000), "middle": (m // 2, m // 2 + 5000)}[where]
        src = rng.integers(lo, hi, 100000)
        dst = rng.integers(lo, hi, 100000)
        if where == "middle":  # and one edge at either end
            src = np.concatenate([src, [0, m - 1]])
            dst = np.concatenate([dst, [m - 1, 0]])
    for sym in (False, True):
        got = graphio.build_csr_device(m, src, dst, sym)
        s, d = (np.concatenate([src, dst]), np.concatenate([dst, src])) if sym else (src, dst)
        assert same(got, graphio.build_csr(m, s, d)), (where, sym)


@pytest.mark.gpu
def test_device_builder_rejects_bad_ids():
    with pytest.raises(_cabi.GardeniaError) as ei:
        graphio.build_csr_device(4, np.array([0, 4]), np.array([1, 2]))
    assert ei.value.status == _cabi.GDN_ERR_INVALID
    with pytest.raises(_cabi.GardeniaError):
        graphio.build_csr_device(4, np.array([0, 1]), np.array([-1, 2]))


@pytest.mark.gpu
def test_mtx2bin_device_build_equals_host_build(tmp_path):
    for name, sym in (("test_bc", 1), ("test_cc", 1), ("chesapeake", 1), ("4", 0), ("test_pr", 0)):
        rc, out = run("mtx2bin", os.path.join(G, name), tmp_path / (name + "_d"), sym)
        assert rc == 0, out
        rc, out2 = run("mtx2bin", os.path.join(G, name), tmp_path / (name + "_h"), sym, 1)
        assert rc == 0, out2
        assert same(graphio.read_bin(str(tmp_path / (name + "_d"))), graphio.read_bin(str(tmp_path / (name + "_h"))))
        for ext in (".meta.txt", ".vertex.bin", ".edge.bin"):
            assert open(str(tmp_path / (name + "_d")) + ext, "rb").read() == open(str(tmp_path / (name + "_h")) + ext, "rb").read()
    # and the converted graph feeds a kernel main
    rc, out = run("mtx2bin", os.path.join(G, "chesapeake"), tmp_path / "ch", 1)
    rc, out = run("tc_hip", tmp_path / "ch")
    assert rc == 0 and "total_num_triangles = 194" in out, out


@pytest.mark.gpu
@pytest.mark.parametrize("scale,ef,seed", [(8, 4, 1), (14, 16, 2), (17, 16, 3)])
def test_device_symmetrize_matches_numpy(scale, ef, seed):
    import ctypes as C
    L = _cabi.lib()
    g = graphio.rmat_graph(scale, ef, seed=seed)
    h, hs = C.c_void_p(), C.c_void_p()
    _cabi.check(L.gdn_graph_upload(g.m, g.nnz, g.rowptr.ctypes.data_as(C.c_void_p), g.colidx.ctypes.data_as(C.c_void_p),
                                   C.byref(h)))
    _cabi.check(L.gdn_graph_symmetrize(h, C.byref(hs)))
    m, nnz = C.c_int32(), C.c_uint64()
    _cabi.check(L.gdn_graph_info(hs, C.byref(m), C.byref(nnz), None, None))
    rp, ci = np.empty(g.m + 1, np.uint64), np.empty(nnz.value, np.int32)
    _cabi.check(L.gdn_graph_download(hs, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p)))
    L.gdn_graph_free(h)
    L.gdn_graph_free(hs)
    assert same(graphio.CSR(g.m, rp, ci), graphio.symmetrize(g))
