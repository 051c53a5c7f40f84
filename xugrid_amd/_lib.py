"""
ctypes binding of ``libxugrid_amd.so`` (the C ABI declared in ``include/xugrid_amd.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C xugrid_amd/csrc``.  There
is no CPU fallback: if the shared object is missing, or no HIP device is present, every compute
entry point raises -- loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# XUGRID_AMD_LIB: measurement hook only -- an A/B build of the same sources with other compiler flags (profiles/fma_ab.sh)
LIB_PATH = os.environ.get("XUGRID_AMD_LIB") or os.path.join(_HERE, "libxugrid_amd.so")

XR_OK = 0
XR_ERR_INVALID = -1
XR_ERR_NO_DEVICE = -2
XR_ERR_HIP = -3
XR_ERR_LIMIT = -4

XR_F64 = 0
XR_F32 = 1


class XugridAmdError(RuntimeError):
    """HIP runtime failure, missing device or missing shared library."""


c_i64 = ctypes.c_int64
c_f64 = ctypes.c_double
c_int = ctypes.c_int
vp = ctypes.c_void_p
p_i64 = ctypes.POINTER(ctypes.c_int64)
p_f64 = ctypes.POINTER(ctypes.c_double)
p_int = ctypes.POINTER(ctypes.c_int)
p_vp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes).  Kept in one table so tests can check that every symbol declared
# in include/xugrid_amd.h is exported and bound.
SIGNATURES = {
    "xr_last_error": (ctypes.c_char_p, []),
    "xr_device_count": (c_int, [p_int]),
    "xr_init": (c_int, [c_int]),
    "xr_current_device": (c_int, [p_int]),
    "xr_trim_pool": (c_int, []),
    "xr_version": (c_int, []),
    "xr_set_stream": (c_int, [vp, c_int, c_int]),
    "xr_set_async": (c_int, [c_int]),
    "xr_set_option": (c_int, [ctypes.c_char_p, ctypes.c_int64]),
    "xr_get_option": (c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "xr_host_copy": (c_int, [vp, vp, c_i64]),
    "xr_host_interleave2": (c_int, [vp, c_i64, vp, c_i64, c_i64, vp]),
    "xr_mesh_create": (c_int, [vp, c_i64, vp, c_int, c_i64, c_i64, c_i64, p_vp]),
    "xr_mesh_create_dev": (c_int, [vp, c_i64, vp, c_int, c_i64, c_i64, c_i64, p_vp]),
    "xr_mesh_create_rectilinear": (c_int, [vp, c_i64, vp, c_i64, p_vp]),
    "xr_mesh_destroy": (c_int, [vp]),
    "xr_mesh_info": (c_int, [vp, p_i64, p_i64, p_i64]),
    "xr_mesh_device_bytes": (c_int, [vp, p_i64]),
    "xr_mesh_prepare": (c_int, [vp]),
    "xr_mesh_build_index": (c_int, [vp]),
    "xr_mesh_invalidate": (c_int, [vp]),
    "xr_mesh_area": (c_int, [vp, vp]),
    "xr_mesh_centroids": (c_int, [vp, vp]),
    "xr_mesh_faces": (c_int, [vp, vp]),
    "xr_mesh_download": (c_int, [vp, vp, vp]),
    "xr_voronoi_create": (c_int, [vp, p_vp]),
    "xr_voronoi_info": (c_int, [vp, p_i64, p_i64, p_i64, p_i64, p_i64]),
    "xr_voronoi_download": (c_int, [vp, vp, vp, vp, vp, vp]),
    "xr_voronoi_boundary_info": (c_int, [vp, vp, vp]),
    "xr_voronoi_boundary": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "xr_voronoi_mesh": (c_int, [vp, vp, c_i64, vp, c_i64, c_i64, p_vp]),
    "xr_voronoi_mesh_auto": (c_int, [vp, p_vp, p_i64, p_i64]),
    "xr_voronoi_tail": (c_int, [vp, vp, vp]),
    "xr_voronoi_boundary_cells_info": (c_int, [vp, p_i64, p_i64, p_i64]),
    "xr_voronoi_boundary_cells": (c_int, [vp, vp, vp]),
    "xr_voronoi_destroy": (c_int, [vp]),
    "xr_overlap": (c_int, [vp, vp, c_int, p_vp]),
    "xr_overlap_stats": (c_int, [vp, p_i64]),
    "xr_overlap_apply_dev": (c_int, [vp, vp, c_int, c_int, c_f64, vp, c_int, c_i64, vp, p_vp]),
    "xr_overlap_partial_dev": (c_int, [vp, vp, c_int, c_int, vp, c_int, c_i64, vp, c_int, p_vp]),
    "xr_shard_plan_dev": (c_int, [vp, vp, c_i64, c_int, vp, vp, c_i64, c_int, c_int, c_int, c_int, vp, p_i64, vp, p_i64, vp]),
    "xr_locate_points": (c_int, [vp, vp, c_i64, c_f64, vp]),
    "xr_locate_raster": (c_int, [vp, vp, c_i64, vp, c_i64, c_f64, vp]),
    "xr_barycentric": (c_int, [vp, vp, c_i64, c_f64, vp, vp]),
    "xr_locate_csr": (c_int, [vp, vp, vp, c_i64, c_f64, p_vp]),
    "xr_replace_interpolated_weights": (c_int, [vp, c_i64, vp, c_i64, c_i64, vp, vp, c_i64, vp, c_i64]),
    "xr_barycentric_csr": (c_int, [vp, vp, vp, vp, c_i64, c_f64, vp, vp, c_i64, p_vp]),
    "xr_barycentric_csr_tail": (c_int, [vp, vp, vp, vp, c_i64, c_f64, c_i64, vp, vp, c_i64, c_int, p_vp]),
    "xr_locate_flags_begin": (c_int, [vp, vp, vp, c_i64, p_vp]),
    "xr_points_destroy": (c_int, [vp]),
    "xr_barycentric_csr_points": (c_int, [vp, vp, vp, c_f64, c_i64, vp, vp, c_i64, c_int, p_vp]),
    "xr_csr_info": (c_int, [vp, p_i64, p_i64, p_i64]),
    "xr_csr_download": (c_int, [vp, vp, vp, vp]),
    "xr_csr_upload": (c_int, [vp, vp, vp, c_i64, c_i64, c_i64, p_vp]),
    "xr_csr_from_triplet": (c_int, [vp, vp, vp, c_i64, c_i64, c_i64, p_vp]),
    "xr_csr_from_outer": (c_int, [vp, vp, vp, c_i64, c_i64, vp, vp, vp, c_i64, c_i64, p_vp]),
    "xr_outer_create": (c_int, [vp, vp, vp, c_i64, c_i64, vp, vp, vp, c_i64, c_i64, p_vp]),
    "xr_outer_info": (c_int, [vp, vp, vp, vp]),
    "xr_outer_csr": (c_int, [vp, p_vp]),
    "xr_outer_destroy": (c_int, [vp]),
    "xr_apply_outer": (c_int, [vp, c_int, c_f64, vp, c_int, c_i64, vp]),
    "xr_apply_outer_dev": (c_int, [vp, c_int, c_f64, vp, c_int, c_i64, vp]),
    "xr_edge_length_csr": (c_int, [vp, vp, c_i64, p_vp]),
    "xr_edge_length_csr_dev": (c_int, [vp, vp, c_i64, p_vp]),
    "xr_edge_pieces": (c_int, [vp, vp, vp, c_i64, vp]),
    "xr_csr_set_row_keys": (c_int, [vp, vp, c_i64]),
    "xr_csr_set_col_keys": (c_int, [vp, vp, c_i64]),
    "xr_csr_col_order": (c_int, [vp, vp]),
    "xr_csr_expect_permuted": (c_int, [vp, c_int]),
    "xr_csr_output_stored_order": (c_int, [vp, c_int]),
    "xr_csr_row_order": (c_int, [vp, c_i64, vp]),
    "xr_csr_destroy": (c_int, [vp]),
    "xr_apply_csr": (c_int, [vp, c_int, c_f64, vp, c_int, c_i64, vp]),
    "xr_apply_csr_dev": (c_int, [vp, c_int, c_f64, vp, c_int, c_i64, vp]),
    "xr_apply_coo": (c_int, [vp, vp, c_i64, c_i64, vp, c_int, c_i64, c_i64, vp]),
    "xr_partial_components": (c_int, [c_int]),
    "xr_partial_combine_is_max": (c_int, [c_int]),
    "xr_apply_partial_dev": (c_int, [vp, c_int, vp, c_int, c_i64, vp, c_int]),
    "xr_partial_fill_identity_dev": (c_int, [c_int, vp, c_i64, c_i64]),
    "xr_finalize_partial_dev": (c_int, [c_int, vp, c_i64, c_i64, vp]),
    "xr_reduce_partial_rows_dev": (c_int, [c_int, vp, vp, vp, c_i64, c_i64, vp]),
    "xr_dev_alloc": (c_int, [c_i64, p_vp]),
    "xr_dev_free": (c_int, [vp]),
    "xr_dev_upload": (c_int, [vp, vp, c_i64]),
    "xr_dev_download": (c_int, [vp, vp, c_i64]),
    "xr_dev_sync": (c_int, []),
    "xr_prof_enable": (c_int, [c_int]),
    "xr_prof_reset": (c_int, []),
    "xr_prof_count": (c_int, [p_int]),
    "xr_prof_get": (c_int, [c_int, ctypes.c_char_p, c_int, p_i64, p_f64]),
}

_lib = None


def load():
    """Load the shared library (no device needed for this step) and bind every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XugridAmdError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C xugrid_amd/csrc). "
            "xugrid_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error():
    msg = load().xr_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Map a C status to the exception type the reference would raise for the same mistake."""
    if rc == XR_OK:
        return
    msg = last_error()
    if rc == XR_ERR_INVALID:
        raise ValueError(msg)
    if rc == XR_ERR_LIMIT:
        raise OverflowError(msg)
    raise XugridAmdError(msg or f"libxugrid_amd error {rc}")


def device_count():
    n = c_int(0)
    load().xr_device_count(ctypes.byref(n))
    return n.value
