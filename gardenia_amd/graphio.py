"""Host-side graph ingest for the CSR hot path (numpy).

Restates the *semantics* of the reference loader ``include/csr_graph.h``:

* ``read_mtx``        -- csr_graph.h:74-120: skip ``%`` header lines, first remaining line is
                         ``m n nnz``; every following non-empty, non-``#`` line contributes
                         ``src dst`` (1-based, extra columns ignored); self loops dropped
                         (:108); optional symmetrize (:113-116).
* ``build_csr``       -- csr_graph.h:122-169 ``fill_data``: neighbour lists sorted ascending,
                         duplicates removed, 64-bit row offsets.
* ``transpose``       -- csr_graph.h:170-194 ``build_reverse_graph``.
* ``read_bin``/``write_bin`` -- the ``<prefix>.meta.txt / .vertex.bin / .edge.bin`` layout read by
                         csr_graph.h:219-230 and src/common/graph.cc:4-18 (the reference ships
                         no writer for it).
* ``orient_dag``      -- src/common/graph.cc:67-113 (degree-then-id orientation used by TC).
* ``rmat_edges``      -- Graph500 R-MAT recipe of include/generator.h:81-114 (A=.57 B=.19
                         C=.19, seed constant 27491095 from include/misc.h:18) on a
                         counter-based integer RNG that the device generator
                         (csrc/rmat.hip) reproduces bit-for-bit.

Layout everywhere: ``rowptr`` = (m+1) x uint64, ``colidx`` = nnz x int32 ascending per row.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

K_RAND_SEED = 27491095  # include/misc.h:18 kRandSeed

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)

# R-MAT quadrant thresholds on a 32-bit draw (A=.57, A+B=.76, A+B+C=.95), include/generator.h:83.
RMAT_TA = int(0.57 * 2**32)
RMAT_TAB = int(0.76 * 2**32)
RMAT_TABC = int(0.95 * 2**32)


@dataclass
class CSR:
    """A CSR graph on raw arrays (the G1 raw-array contract with G2 widths)."""

    m: int
    rowptr: np.ndarray  # uint64, m+1
    colidx: np.ndarray  # int32, nnz

    @property
    def nnz(self) -> int:
        return int(self.rowptr[-1])

    def degrees(self) -> np.ndarray:
        return np.diff(self.rowptr.astype(np.int64)).astype(np.int32)


def build_csr(m: int, src: np.ndarray, dst: np.ndarray, dedupe: bool = True,
              drop_self_loops: bool = True) -> CSR:
    """Sort rows ascending and drop duplicates (csr_graph.h:122-169)."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    if drop_self_loops:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    key = src * np.int64(m) + dst
    key = np.unique(key) if dedupe else np.sort(key, kind="stable")
    s = key // m
    d = (key - s * m).astype(np.int32)
    counts = np.bincount(s, minlength=m).astype(np.uint64)
    rowptr = np.zeros(m + 1, dtype=np.uint64)
    np.cumsum(counts, out=rowptr[1:])
    return CSR(m, rowptr, d)


def build_csr_device(m: int, src: np.ndarray, dst: np.ndarray, symmetrize_: bool = False) -> CSR:
    """The same clean-up as build_csr (+ optional symmetrization) done on the device through
    gdn_graph_from_edges: one radix sort instead of the reference's per-row sort + erase loop
    (csr_graph.h:122-143).  Needs a HIP device (no CPU fallback)."""
    import ctypes as C
    from . import _cabi
    L = _cabi.lib()
    s32 = np.ascontiguousarray(src, dtype=np.int32)
    d32 = np.ascontiguousarray(dst, dtype=np.int32)
    if s32.shape != d32.shape:
        raise ValueError("src and dst must have the same length")
    h = C.c_void_p()
    _cabi.check(L.gdn_graph_from_edges(m, s32.size, s32.ctypes.data_as(C.c_void_p), d32.ctypes.data_as(C.c_void_p),
                                       1 if symmetrize_ else 0, C.byref(h)))
    try:
        mm, nnz = C.c_int32(), C.c_uint64()
        _cabi.check(L.gdn_graph_info(h, C.byref(mm), C.byref(nnz), None, None))
        rowptr = np.empty(m + 1, np.uint64)
        colidx = np.empty(nnz.value, np.int32)
        _cabi.check(L.gdn_graph_download(h, rowptr.ctypes.data_as(C.c_void_p), colidx.ctypes.data_as(C.c_void_p)))
    finally:
        L.gdn_graph_free(h)
    return CSR(m, rowptr, colidx)


def csr_to_coo(g: CSR) -> Tuple[np.ndarray, np.ndarray]:
    src = np.repeat(np.arange(g.m, dtype=np.int64), np.diff(g.rowptr.astype(np.int64)))
    return src, g.colidx.astype(np.int64)


def transpose(g: CSR) -> CSR:
    """Reverse graph (csr_graph.h:170-194): in-neighbour lists come out ascending."""
    src, dst = csr_to_coo(g)
    return build_csr(g.m, dst, src, dedupe=False, drop_self_loops=False)


def symmetrize(g: CSR) -> CSR:
    src, dst = csr_to_coo(g)
    return build_csr(g.m, np.concatenate([src, dst]), np.concatenate([dst, src]))


def read_mtx(path: str, symmetrize_: bool = False) -> CSR:
    """csr_graph.h:74-120 semantics (banner optional; weights ignored)."""
    with open(path, "r") as f:
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        parts = line.split()
        m = int(parts[0])
        srcs, dsts = [], []
        for line in f:
            if len(line.strip()) == 0 or line[0] == "#":
                continue
            p = line.split()
            if len(p) < 2:
                break  # next_line(): a failed parse ends the read loop
            a, b = int(p[0]), int(p[1])
            if a == b:
                continue
            srcs.append(a - 1)
            dsts.append(b - 1)
    src = np.asarray(srcs, dtype=np.int64)
    dst = np.asarray(dsts, dtype=np.int64)
    if symmetrize_:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    return build_csr(m, src, dst)


def write_mtx(path: str, g: CSR) -> None:
    src, dst = csr_to_coo(g)
    with open(path, "w") as f:
        f.write(f"{g.m} {g.m} {g.nnz}\n")
        for a, b in zip(src.tolist(), dst.tolist()):
            f.write(f"{a + 1} {b + 1}\n")


def write_bin(prefix: str, g: CSR) -> None:
    """<prefix>.meta.txt = 'n_vertices n_edges sizeof(vid) max_degree'; .vertex.bin =
    (n+1) x 64-bit offsets; .edge.bin = nnz x int32 (csr_graph.h:219-230)."""
    deg = g.degrees()
    with open(prefix + ".meta.txt", "w") as f:
        f.write(f"{g.m}\n{g.nnz}\n4\n{int(deg.max()) if g.m else 0}\n")
    g.rowptr.astype(np.uint64).tofile(prefix + ".vertex.bin")
    g.colidx.astype(np.int32).tofile(prefix + ".edge.bin")


def read_bin(prefix: str) -> CSR:
    with open(prefix + ".meta.txt") as f:
        vals = f.read().split()
    m, nnz, vid_size = int(vals[0]), int(vals[1]), int(vals[2])
    if vid_size != 4:
        raise ValueError("vertex id size must be 4 (common.h:35 vidType=int32)")
    rowptr = np.fromfile(prefix + ".vertex.bin", dtype=np.uint64, count=m + 1)
    colidx = np.fromfile(prefix + ".edge.bin", dtype=np.int32, count=nnz)
    return CSR(m, rowptr, colidx)


def orient_dag(g: CSR) -> CSR:
    """Keep u->v iff deg[v] > deg[u] or (deg equal and v > u): src/common/graph.cc:80-81."""
    deg = np.diff(g.rowptr.astype(np.int64))
    src, dst = csr_to_coo(g)
    keep = (deg[dst] > deg[src]) | ((deg[dst] == deg[src]) & (dst > src))
    return build_csr(g.m, src[keep], dst[keep], dedupe=False, drop_self_loops=False)


# --------------------------------------------------------------------------------------
# R-MAT (Graph500 recipe, counter-based RNG shared with the device generator)
# --------------------------------------------------------------------------------------

def _mix64(z: np.ndarray) -> np.ndarray:
    z = z.copy()
    z ^= z >> np.uint64(30)
    z *= _M1
    z ^= z >> np.uint64(27)
    z *= _M2
    z ^= z >> np.uint64(31)
    return z


def permute_id(v: np.ndarray, scale: int, seed: int) -> np.ndarray:
    """Bijection on [0, 2^scale) standing in for generator.h:52-62 PermuteIDs (every step is
    invertible on scale-bit integers: odd multiply, xor-shift-right, add)."""
    mask = np.uint64((1 << scale) - 1)
    half = np.uint64(max(1, scale // 2))
    s = np.uint64(seed)
    x = v.astype(np.uint64) & mask
    with np.errstate(over="ignore"):
        x = (x * np.uint64(0x9E3779B1) + s) & mask
        x ^= x >> half
        x = (x * np.uint64(0x85EBCA6B) + np.uint64(0xC2B2AE35)) & mask
        x ^= x >> half
        x = (x * np.uint64(0x27D4EB2F) + np.uint64(0x165667B1)) & mask
        x ^= x >> half
    return x


def rmat_thresholds(a: float = 0.57, b: float = 0.19, c: float = 0.19) -> Tuple[int, int, int]:
    """floor(p * 2^32) of the running quadrant sums, as gdn_rmat_build_ex computes them (doubles); (.57, .19, .19) give
    RMAT_TA / RMAT_TAB / RMAT_TABC."""
    thr = lambda p: 0xFFFFFFFF if p >= 1.0 else int(float(p) * 4294967296.0)
    return thr(a), thr(float(a) + float(b)), thr(float(a) + float(b) + float(c))


def rmat_edges(scale: int, edge_factor: int = 16, seed: int = K_RAND_SEED,
               permute: bool = True, lo: int = 0, hi: Optional[int] = None,
               abc: Optional[Tuple[float, float, float]] = None, n_edges: Optional[int] = None
               ) -> Tuple[np.ndarray, np.ndarray]:
    """Edges [lo,hi) of the R-MAT stream (src,dst as int64).  Edge e, level l draws
    r = hi32/lo32 of mix64(seed + e*GOLD + (l>>1)*M1); r<TA: (0,0); <TAB: dst bit;
    <TABC: src bit; else both (include/generator.h:94-106).  abc: other quadrant probabilities (gdn_rmat_build_ex);
    n_edges: the stream's length when it is not edge_factor * 2^scale."""
    total = (edge_factor << scale) if n_edges is None else int(n_edges)
    hi = total if hi is None else hi
    ta, tab, tabc = (RMAT_TA, RMAT_TAB, RMAT_TABC) if abc is None else rmat_thresholds(*abc)
    e = np.arange(lo, hi, dtype=np.uint64)
    src = np.zeros(e.shape, dtype=np.uint64)
    dst = np.zeros(e.shape, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = np.uint64(seed) + e * _GOLD
        for l in range(scale):
            if (l & 1) == 0:
                h = _mix64(base + np.uint64(l >> 1) * _M1)
                r = h & np.uint64(0xFFFFFFFF)
            else:
                r = h >> np.uint64(32)
            src <<= np.uint64(1)
            dst <<= np.uint64(1)
            dst |= ((r >= np.uint64(ta)) & (r < np.uint64(tab))).astype(np.uint64)
            src |= ((r >= np.uint64(tab)) & (r < np.uint64(tabc))).astype(np.uint64)
            both = (r >= np.uint64(tabc)).astype(np.uint64)
            src |= both
            dst |= both
    if permute:
        src = permute_id(src, scale, seed)
        dst = permute_id(dst, scale, seed)
    return src.astype(np.int64), dst.astype(np.int64)


def rmat_graph_ex(scale: int, n_edges: int, abc: Tuple[float, float, float] = (0.57, 0.19, 0.19), seed: int = K_RAND_SEED,
                  permute: bool = True, compact: bool = False) -> CSR:
    """numpy twin of gdn_rmat_build_ex (include/gardenia_hip.h).  compact: the ids that occur in no edge (self loops do not
    count) are dropped and the others renumbered in ascending order."""
    src, dst = rmat_edges(scale, 0, seed, permute, abc=abc, n_edges=n_edges)
    m = 1 << scale
    if compact:
        real = src != dst
        live = np.zeros(m, bool)
        live[src[real]] = True
        live[dst[real]] = True
        new_id = np.cumsum(live) - 1
        src, dst = new_id[src[real]], new_id[dst[real]]
        m = max(int(live.sum()), 1)
    return build_csr(m, src, dst)


# ---- stand-ins for the real graphs of BASELINE configs 2 and 4 (datasets/test.mk:5,8 are wget lines; no network here):
# seeded, generated on the device by gdn_rmat_build_ex with the same arguments.  Recipes fitted in round 5 to the published
# sizes (SNAP): soc-LiveJournal1 4 847 571 vertices / 68 993 773 directed edges, max in-degree 13 906, out 20 293;
# com-Orkut 3 072 441 vertices / 117 185 083 undirected edges, max degree 33 313.  What comes out is in DESIGN.md 6.
LJ_LIKE = dict(scale=23, n_edges=70_000_000, abc=(0.5, 0.2, 0.2), flags=3)          # directed
ORKUT_LIKE = dict(scale=22, n_edges=118_000_000, abc=(0.45, 0.22, 0.22), flags=3)   # then symmetrized


def standin_graph(recipe: dict, seed: int = K_RAND_SEED) -> CSR:
    return rmat_graph_ex(recipe["scale"], recipe["n_edges"], recipe["abc"], seed, bool(recipe["flags"] & 1),
                         bool(recipe["flags"] & 2))


def rmat_graph(scale: int, edge_factor: int = 16, seed: int = K_RAND_SEED,
               permute: bool = True) -> CSR:
    """Directed R-MAT graph cleaned like the reference loader (self loops and duplicates
    dropped, rows ascending)."""
    src, dst = rmat_edges(scale, edge_factor, seed, permute)
    return build_csr(1 << scale, src, dst)


# ---- non-R-MAT shapes (SURVEY 8d stand-ins for real graphs: the heuristics of the solvers -- tier picker, level / bucket
# choosers, dense-sweep triggers -- were tuned on R-MAT; these exercise the other regimes).  Edge lists, numpy, seeded.
def grid2d_edges(nx: int, ny: int) -> Tuple[int, np.ndarray, np.ndarray]:
    """Road-like: nx x ny lattice, 4 neighbours, both directions; diameter nx + ny, every degree <= 4."""
    idx = np.arange(nx * ny, dtype=np.int64).reshape(ny, nx)
    a = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()])
    b = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()])
    return nx * ny, np.concatenate([a, b]), np.concatenate([b, a])


def uniform_edges(m: int, n_edges: int, seed: int = 1) -> Tuple[int, np.ndarray, np.ndarray]:
    """Erdos-Renyi-like: endpoints uniform; no hubs at all (Poisson degrees)."""
    rng = np.random.default_rng(seed)
    return m, rng.integers(0, m, n_edges, dtype=np.int64), rng.integers(0, m, n_edges, dtype=np.int64)


def small_world_edges(m: int, k: int, p: float, seed: int = 1) -> Tuple[int, np.ndarray, np.ndarray]:
    """Watts-Strogatz: ring lattice of the k nearest neighbours (both directions), each far end rewired with probability
    p: clustered like a social graph (many triangles), low diameter, narrow degree distribution."""
    rng = np.random.default_rng(seed)
    src = np.repeat(np.arange(m, dtype=np.int64), k // 2)
    dst = (src + np.tile(np.arange(1, k // 2 + 1, dtype=np.int64), m)) % m
    rew = rng.random(src.size) < p
    dst = np.where(rew, rng.integers(0, m, src.size, dtype=np.int64), dst)
    return m, np.concatenate([src, dst]), np.concatenate([dst, src])


def first_nonisolated(g: CSR) -> int:
    deg = g.degrees()
    nz = np.nonzero(deg)[0]
    return int(nz[0]) if len(nz) else 0


def reference_dataset(name: str) -> str:
    """Path of a small fixture graph copied (as data) from the reference's datasets/ and
    test/graphs/ into tests/golden/graphs/."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.path.join(here, "tests", "golden", "graphs", name)
