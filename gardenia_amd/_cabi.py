"""ctypes binding of gardenia_amd/lib/libgardenia_hip.so (the C-ABI of include/gardenia_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at import of
the first symbol, and every entry point returns GDN_ERR_NO_DEVICE without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GARDENIA_HIP_LIB: another build of the same library (tools/ A/B measurements of compile-time variants)
LIB_PATH = os.environ.get("GARDENIA_HIP_LIB") or os.path.join(_HERE, "lib", "libgardenia_hip.so")

GDN_OK = 0
GDN_ERR_INVALID = -1
GDN_ERR_NO_DEVICE = -2
GDN_ERR_HIP = -3
GDN_ERR_OOM = -4
GDN_ERR_OVERFLOW = -5
GDN_LAYOUT_AUTO, GDN_LAYOUT_CSR, GDN_LAYOUT_PB, GDN_LAYOUT_PB_SQUISHED = -1, 0, 1, 2
GDN_PR_PART_FIRST, GDN_PR_PART_LAST = 1, 2


class GdnStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("reserved", C.c_int32), ("solve_ms", C.c_double),
                ("h2d_ms", C.c_double), ("prep_ms", C.c_double), ("edges_traversed", C.c_uint64),
                ("last_error", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class GardeniaError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"libgardenia_hip status {status}: {msg}")
        self.status = status


_vp = C.c_void_p
_i32 = C.c_int32
_u64 = C.c_uint64
_pp = C.POINTER(C.c_void_p)
_st = C.POINTER(GdnStats)

# name -> (restype, argtypes); must list every symbol of include/gardenia_hip.h
PROTOTYPES = {
    "gdn_last_error": (C.c_char_p, []),
    "gdn_option_set": (C.c_int, [C.c_char_p, C.c_char_p]),
    "gdn_option_get": (C.c_int, [C.c_char_p, C.c_char_p, _i32]),
    "gdn_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gdn_set_device": (C.c_int, [C.c_int]),
    "gdn_bfs": (C.c_int, [_i32, _u64, _vp, _vp, _vp, _vp, _i32, _vp, _st]),
    "gdn_pr": (C.c_int, [_i32, _u64, _vp, _vp, _vp, _vp, C.c_float, C.c_double, _i32, _st]),
    "gdn_pr_delta": (C.c_int, [_i32, _u64, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_double, C.c_float, _i32, _i32, _st]),
    "gdn_pr_delta_plan_create": (C.c_int, [_vp, _vp, _i32, _pp]),
    "gdn_pr_delta_run": (C.c_int, [_vp, _vp, C.c_float, C.c_double, C.c_float, _i32, _i32, _st]),
    "gdn_pr_delta_trace": (C.c_int, [_vp, _i32, C.POINTER(_i32), _vp, _vp, _vp]),
    "gdn_pr_delta_plan_free": (C.c_int, [_vp]),
    "gdn_pr_last_trace": (C.c_int, [_i32, C.POINTER(_i32), _vp]),
    "gdn_pr_last_layout": (C.c_int, [C.POINTER(_i32)]),
    "gdn_pr_multi": (C.c_int, [_i32, _u64, _vp, _vp, _vp, _vp, C.c_float, C.c_double, _i32, _i32, _vp, _st]),
    "gdn_spmv_multi": (C.c_int, [_i32, _u64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _st]),
    "gdn_multi_ranges": (C.c_int, [_i32, _vp, _i32, _vp, C.POINTER(_i32)]),
    "gdn_graph_upload_rows": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _i32, _pp]),
    "gdn_graph_validate": (C.c_int, [_vp, _i32]),
    "gdn_graph_balanced_ranges": (C.c_int, [_vp, _i32, _vp]),
    "gdn_graph_slice_padded": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _pp]),
    "gdn_graph_pad_columns": (C.c_int, [_vp, _i32, _vp, _i32]),
    "gdn_spmv": (C.c_int, [_i32, _u64, _vp, _vp, _vp, _vp, _vp, _st]),
    "gdn_sssp": (C.c_int, [_i32, _u64, _vp, _vp, _vp, _i32, _i32, _vp, _st]),
    "gdn_tc": (C.c_int, [_i32, _u64, _vp, _vp, _i32, C.POINTER(_u64), _st]),
    "gdn_bc": (C.c_int, [_i32, _u64, _vp, _vp, _i32, _vp, C.POINTER(GdnStats)]),
    "gdn_bc_dev": (C.c_int, [_vp, _i32, _vp, C.POINTER(GdnStats)]),
    "gdn_bc_plan_create": (C.c_int, [_vp, _vp, _pp]),
    "gdn_bc_run": (C.c_int, [_vp, _i32, _vp, C.POINTER(GdnStats)]),
    "gdn_bc_plan_free": (C.c_int, [_vp]),
    "gdn_cc": (C.c_int, [_i32, _u64, _vp, _vp, _vp, _vp, _vp, _st]),
    "gdn_dev_alloc": (C.c_int, [_u64, _pp]),
    "gdn_dev_free": (C.c_int, [_vp]),
    "gdn_dev_trim": (C.c_int, [C.POINTER(_u64)]),
    "gdn_dev_reserve": (C.c_int, [_u64]),
    "gdn_dev_upload": (C.c_int, [_vp, _vp, _u64]),
    "gdn_dev_download": (C.c_int, [_vp, _vp, _u64]),
    "gdn_sort_u64_dev": (C.c_int, [_vp, _vp, _u64, _i32, _i32, _pp]),
    "gdn_worklist_filter_dev": (C.c_int, [_vp, _i32, _i32, _vp, C.c_uint32, _vp, _vp]),
    "gdn_graph_upload": (C.c_int, [_i32, _u64, _vp, _vp, _pp]),
    "gdn_graph_wrap_dev": (C.c_int, [_i32, _u64, _vp, _vp, _pp]),
    "gdn_graph_free": (C.c_int, [_vp]),
    "gdn_graph_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_u64), _pp, _pp]),
    "gdn_graph_degrees_dev": (C.c_int, [_vp, _vp, _vp]),
    "gdn_graph_transpose": (C.c_int, [_vp, _pp]),
    "gdn_graph_symmetrize": (C.c_int, [_vp, _pp]),
    "gdn_graph_slice_rows": (C.c_int, [_vp, _i32, _i32, _pp]),
    "gdn_graph_from_edges": (C.c_int, [_i32, _u64, _vp, _vp, _i32, _pp]),
    "gdn_graph_download": (C.c_int, [_vp, _vp, _vp]),
    "gdn_rmat_build": (C.c_int, [_i32, _i32, _u64, _i32, _pp, _pp]),
    "gdn_rmat_build_ex": (C.c_int, [_i32, _u64, C.c_double, C.c_double, C.c_double, _u64, _i32, _pp, _pp]),
    "gdn_rmat_build_range": (C.c_int, [_i32, _u64, C.c_double, C.c_double, C.c_double, _u64, _i32, _i32, _i32, _pp, _vp]),
    "gdn_pr_plan_create": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _pp]),
    "gdn_pr_plan_layout": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "gdn_pr_plan_hubs": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_u64)]),
    "gdn_pr_plan_mid": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_u64)]),
    "gdn_pr_plan_move": (C.c_int, [_vp, C.c_uint32]),
    "gdn_pr_plan_state_size": (C.c_int, [_vp, C.POINTER(_i32)]),
    "gdn_pr_plan_bins": (C.c_int, [_vp, C.POINTER(_i32)]),
    "gdn_pr_import_dev": (C.c_int, [_vp, _vp, _vp, C.c_float, _vp]),
    "gdn_pr_import_diff": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "gdn_pr_export_dev": (C.c_int, [_vp, _vp, _vp, C.c_float, _vp]),
    "gdn_pr_squish_create": (C.c_int, [_vp, _vp, _pp]),
    "gdn_pr_squish_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), _pp, _pp]),
    "gdn_pr_squish_import_dev": (C.c_int, [_vp, _vp, _vp, C.c_float, C.POINTER(C.c_double), _vp]),
    "gdn_pr_squish_export_dev": (C.c_int, [_vp, _vp, _vp, C.c_float, _vp]),
    "gdn_pr_squish_free": (C.c_int, [_vp]),
    "gdn_pr_squish_degrees_dev": (C.c_int, [_vp, _vp, _vp]),
    "gdn_pr_squish_range": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    "gdn_pr_plan_set_base": (C.c_int, [_vp, _i32]),
    "gdn_pr_plan_check": (C.c_int, [_vp]),
    "gdn_pr_plan_free": (C.c_int, [_vp]),
    "gdn_pr_contrib_dev": (C.c_int, [_vp, _vp, _vp, _vp]),
    "gdn_pr_pull_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_float, _vp]),
    "gdn_pr_pull_rows_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_float, _i32, _i32, _i32, _vp]),
    "gdn_pr_plan_refsum_info": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "gdn_pr_pull_parts_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_float, _i32, _vp, _vp]),
    "gdn_pr_wait_part_dev": (C.c_int, [_vp, _i32, _vp]),
    "gdn_pr_plan_kernel_time": (C.c_int, [_vp, _i32, _i32, C.POINTER(C.c_double), C.POINTER(_i32)]),
    "gdn_pr_iter_bytes": (_u64, [_vp]),
    "gdn_spmv_plan_create": (C.c_int, [_vp, _vp, _i32, _pp]),
    "gdn_spmv_plan_create_cols": (C.c_int, [_vp, _vp, _i32, _i32, _pp]),
    "gdn_spmv_plan_check": (C.c_int, [_vp]),
    "gdn_spmv_plan_tiers": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_u64)]),
    "gdn_spmv_plan_free": (C.c_int, [_vp]),
    "gdn_spmv_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "gdn_spmv_plan_kernel_time": (C.c_int, [_vp, _i32, _i32, C.POINTER(C.c_double), C.POINTER(_i32)]),
    "gdn_spmv_bytes": (_u64, [_vp]),
    "gdn_bfs_dev": (C.c_int, [_vp, _vp, _i32, _vp, _st]),
    "gdn_bfs_plan_create": (C.c_int, [_vp, _vp, _i32, _pp]),
    "gdn_bfs_plan_free": (C.c_int, [_vp]),
    "gdn_bfs_run": (C.c_int, [_vp, _i32, _vp, _st]),
    "gdn_sssp_dev": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _st]),
    "gdn_sssp_plan_create": (C.c_int, [_vp, _vp, _i32, _pp]),
    "gdn_sssp_plan_free": (C.c_int, [_vp]),
    "gdn_sssp_run": (C.c_int, [_vp, _i32, _i32, _vp, _st]),
    "gdn_cc_dev": (C.c_int, [_vp, _vp, _vp, _st]),
    "gdn_tc_dev": (C.c_int, [_vp, _i32, C.POINTER(_u64), _st]),
    "gdn_graph_orient": (C.c_int, [_vp, _pp]),
    "gdn_tc_model_bytes": (C.c_int, [_vp, C.POINTER(_u64)]),
    "gdn_tc_probe_counts": (C.c_int, [_vp, C.POINTER(_u64)]),
    "gdn_tc_plan_create": (C.c_int, [_vp, _i32, C.POINTER(_vp)]),
    "gdn_tc_plan_count": (C.c_int, [_vp, C.POINTER(_u64), _st]),
    "gdn_tc_plan_free": (C.c_int, [_vp]),
    "gdn_tc_plan_walked_elements": (C.c_int, [_vp, C.POINTER(_u64)]),
    "gdn_tc_rows_dev": (C.c_int, [_vp, _i32, _i32, C.POINTER(_u64), _st]),
}

_lib = None


def lib():
    """Load the C-ABI library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C gardenia_amd/csrc).  gardenia_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int):
    if status != GDN_OK:
        msg = lib().gdn_last_error()
        raise GardeniaError(status, msg.decode() if msg else "")
    return status


def device_count() -> int:
    n = C.c_int(0)
    lib().gdn_device_count(C.byref(n))
    return n.value
