"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol that
include/gardenia_hip.h declares, the ctypes table covers the header, and -- without a GPU --
every compute entry point fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from gardenia_amd import _cabi

HEADER = os.path.join(ROOT, "include", "gardenia_hip.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gdn_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 30
    L = _cabi.lib()
    for s in syms:
        assert hasattr(L, s), f"{s} declared in gardenia_hip.h but not exported"
        assert s in _cabi.PROTOTYPES, f"{s} has no ctypes prototype"
    assert sorted(_cabi.PROTOTYPES) == syms


def test_stats_struct_layout_matches_header():
    assert C.sizeof(_cabi.GdnStats) == 48
    assert _cabi.GdnStats.solve_ms.offset == 8 and _cabi.GdnStats.edges_traversed.offset == 32


def test_no_cpu_fallback_without_gpu():
    if _cabi.device_count() > 0:
        pytest.skip("a GPU is present")
    from gardenia_amd import graphio, solvers
    g = solvers.Graph(csr=graphio.rmat_graph(6, 4), need_reverse=True)
    dist = np.full(g.V(), solvers.MYINFINITY, np.int32)
    with pytest.raises(_cabi.GardeniaError) as ei:
        solvers.BFSSolver(g, 0, dist)
    assert ei.value.status == _cabi.GDN_ERR_NO_DEVICE
    scores = np.full(g.V(), 1.0 / g.V(), np.float32)
    with pytest.raises(_cabi.GardeniaError):
        solvers.PRSolver(g, scores)
    assert np.all(dist == solvers.MYINFINITY)  # nothing computed


def test_invalid_arguments_are_reported_not_fatal():
    L = _cabi.lib()
    st = _cabi.GdnStats()
    assert L.gdn_bfs(0, 0, None, None, None, None, 0, None, C.byref(st)) == _cabi.GDN_ERR_INVALID
    assert b"invalid argument" in L.gdn_last_error()
    assert L.gdn_pr(4, 0, None, None, None, None, 0.85, 1e-4, 100, None) == _cabi.GDN_ERR_INVALID


def test_options_api_roundtrip_and_environment_override(monkeypatch):
    """Every GDN_* knob is an option set through the API; the environment variable of the same name overrides it."""
    L = _cabi.lib()
    buf = C.create_string_buffer(64)
    monkeypatch.delenv("GDN_PR_LAYOUT", raising=False)
    assert L.gdn_option_get(b"GDN_PR_LAYOUT", buf, 64) == _cabi.GDN_OK and buf.value == b""
    assert L.gdn_option_set(b"GDN_PR_LAYOUT", b"csr") == _cabi.GDN_OK
    assert L.gdn_option_get(b"GDN_PR_LAYOUT", buf, 64) == _cabi.GDN_OK and buf.value == b"csr"
    monkeypatch.setenv("GDN_PR_LAYOUT", "pb")
    assert L.gdn_option_get(b"GDN_PR_LAYOUT", buf, 64) == _cabi.GDN_OK and buf.value == b"pb"
    monkeypatch.delenv("GDN_PR_LAYOUT")
    assert L.gdn_option_set(b"GDN_PR_LAYOUT", None) == _cabi.GDN_OK
    assert L.gdn_option_get(b"GDN_PR_LAYOUT", buf, 64) == _cabi.GDN_OK and buf.value == b""
    assert L.gdn_option_set(b"PATH", b"x") == _cabi.GDN_ERR_INVALID


def test_option_surface_is_the_documented_public_one():
    """the shipped library reads three classes of options (csrc/gdn_common.hpp).  Every name it reads through
    gdn_option -- the PUBLIC class -- is documented in include/gardenia_hip.h, and there are at most 40 of them; everything else
    goes through gdn_test_option (honoured only under GDN_TEST_HOOKS=1) or gdn_xoption (compiled in only with -DGDN_EXPERIMENTS);
    no name is read through two classes."""
    import glob
    import re
    src = "".join(open(p).read() for p in glob.glob(os.path.join(ROOT, "gardenia_amd", "csrc", "*.h*")))
    header = open(os.path.join(ROOT, "include", "gardenia_hip.h")).read()
    public = set(re.findall(r'gdn_option\("(GDN_[A-Z0-9_]+)"\)', src))
    hooks = set(re.findall(r'gdn_test_option\("(GDN_[A-Z0-9_]+)"\)', src))
    exper = set(re.findall(r'gdn_xoption\("(GDN_[A-Z0-9_]+)"\)', src))
    assert 20 <= len(public) <= 40, sorted(public)
    missing = sorted(n for n in public if n not in header)
    assert not missing, "public options without a line in include/gardenia_hip.h: %s" % missing
    assert not (public & hooks) and not (public & exper) and not (hooks & exper), (public & hooks, public & exper, hooks & exper)
    assert len(hooks) >= 40 and len(exper) >= 40
    # the tests' own hooks are switched on by the conftest, not by the library's defaults
    assert os.environ.get("GDN_TEST_HOOKS") == "1"


def test_tc_hash_set_kernel_leaves_a_simd_room_for_two_core_waves(tmp_path):
    """tc_count_kernel runs four waves per SIMD (LDS) beside tc_core_count_kernel's waves of <= 56 vector registers: at <= 96 registers
    (allocated in blocks of 8) a SIMD's 512 hold two core waves beside them, at 104 one -- RMAT-23 10.8 against 12.7 ms
    (profiles/r06_tc_counters.md section 7).  The code object's own count (.vgpr_count), not the compiler's remark, is what decides."""
    import re
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    hipcc = "/opt/rocm/bin/hipcc"
    if not (os.path.exists(hipcc) and os.path.exists(llvm + "/clang-offload-bundler") and os.path.exists(llvm + "/llvm-readelf")):
        pytest.skip("no ROCm toolchain")
    src = os.path.join(ROOT, "gardenia_amd", "csrc", "gdn_tc.hip")
    obj, co = str(tmp_path / "tc.o"), str(tmp_path / "tc.co")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-c", src, "-o", obj],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    subprocess.run([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--output=" + co], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    notes = subprocess.run([llvm + "/llvm-readelf", "--notes", co], check=True, stdout=subprocess.PIPE, text=True).stdout
    found = {}
    for block in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        vg = re.search(r"\.vgpr_count:\s+(\d+)", block)
        if name and vg:
            found[name.group(1)] = int(vg.group(1))
    count = [v for k, v in found.items() if k.startswith("_Z15tc_count_kernel")]
    core = [v for k, v in found.items() if "tc_core_count_kernel" in k]
    assert count and len(core) == 4, sorted(found)
    assert count[0] <= 96, count
    assert max(core) <= 64, core
    assert 4 * ((count[0] + 7) // 8 * 8) + 2 * ((max(core[:3]) + 7) // 8 * 8) <= 512  # K <= 12288: two waves of the core kernel fit
