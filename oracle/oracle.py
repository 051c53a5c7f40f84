"""
ctypes wrapper around oracle/libxr_oracle.so -- the CPU ORACLE.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and there only as the checker / the timed CPU baseline.  Nothing under
``xugrid_amd/`` imports it (tests/test_layout.py enforces that).

Parity status: see oracle/xr_oracle.h.  The in-tree parts (apply, reducers, CSR, area,
centroids) are pinned against tests/golden/; the polygon clip / search / locate /
barycentric arithmetic restates the absent third-party numba_celltree 0.4.2 and is
"parity unpinned" beyond the reference's own invariants and known answers.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libxr_oracle.so")

METHOD_IDS = {
    "mean": 0,
    "harmonic_mean": 1,
    "geometric_mean": 2,
    "sum": 3,
    "minimum": 4,
    "maximum": 5,
    "mode": 6,
    "percentile": 7,
    "first_order_conservative": 8,
    "conductance": 8,
    "max_overlap": 9,
}


def method_to_id(method):
    """'mean' -> (0, 0.0); 'median' -> (7, 50.0); 'p25' -> (7, 25.0); ('percentile', p)."""
    if isinstance(method, tuple):
        return METHOD_IDS[method[0]], float(method[1])
    if method == "median":
        return 7, 50.0
    if method.startswith("p") and method[1:].isdigit():
        return 7, float(method[1:])
    return METHOD_IDS[method], 0.0


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "xr_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libxr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None

_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)


def _p(a, ty):
    return a.ctypes.data_as(ty)


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.xo_reduce.restype = ctypes.c_double
        L.xo_reduce.argtypes = [ctypes.c_int, ctypes.c_double, _f64p, _f64p, _f64p, ctypes.c_int64]
        L.xo_tree_create.restype = ctypes.c_void_p
        L.xo_tree_create.argtypes = [_f64p, ctypes.c_int64, _i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
        L.xo_tree_destroy.argtypes = [ctypes.c_void_p]
        L.xo_tree_destroy.restype = None
        L.xo_tree_faces.argtypes = [ctypes.c_void_p, _i64p]
        L.xo_clip_area.restype = ctypes.c_double
        L.xo_clip_area.argtypes = [_f64p, ctypes.c_int64, _f64p, ctypes.c_int64]
        L.xo_default_tolerance.restype = ctypes.c_double
        L.xo_default_tolerance.argtypes = [ctypes.c_void_p]
        L.xo_regrid_csr.argtypes = [
            ctypes.c_int, ctypes.c_double, _f64p, ctypes.c_int64, ctypes.c_int64, _f64p, _i64p, _i64p,
            ctypes.c_int64, _f64p, ctypes.c_int,
        ]
        L.xo_regrid_coo.argtypes = [_f64p, ctypes.c_int64, ctypes.c_int64, _i64p, _i64p, ctypes.c_int64, ctypes.c_int64, _f64p]
        L.xo_to_csr_indptr.argtypes = [_i64p, ctypes.c_int64, ctypes.c_int64, _i64p]
        L.xo_area.argtypes = [_f64p, _i64p, ctypes.c_int64, ctypes.c_int64, _f64p]
        L.xo_centroids.argtypes = [_f64p, _i64p, ctypes.c_int64, ctypes.c_int64, _f64p]
        L.xo_replace_interpolated_weights.argtypes = [
            _f64p, _i64p, ctypes.c_int64, _i64p, _f64p, ctypes.c_int64, _i64p, ctypes.c_int64,
        ]
        L.xo_intersect_faces_count.argtypes = [
            ctypes.c_void_p, _f64p, ctypes.c_int64, _i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int, _i64p, _i64p,
        ]
        L.xo_intersect_faces_fill.argtypes = [ctypes.c_void_p, _i64p, _i64p, _f64p]
        L.xo_intersect_faces_bruteforce.argtypes = [
            ctypes.c_void_p, _f64p, ctypes.c_int64, _i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, _i64p, _i64p, _f64p, _i64p,
        ]
        L.xo_intersect_edges_count.argtypes = [ctypes.c_void_p, _f64p, ctypes.c_int64, _i64p]
        L.xo_intersect_edges_fill.argtypes = [ctypes.c_void_p, _i64p, _i64p, _f64p]
        L.xo_locate_points.argtypes = [ctypes.c_void_p, _f64p, ctypes.c_int64, ctypes.c_double, _i64p]
        L.xo_barycentric.argtypes = [ctypes.c_void_p, _f64p, ctypes.c_int64, ctypes.c_double, _i64p, _f64p]
        L.xo_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _xy(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert a.ndim == 2 and a.shape[1] == 2
    return a


def _faces(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    assert a.ndim == 2
    return a


def num_threads():
    return lib().xo_num_threads()


def set_num_threads(n):
    lib().xo_set_num_threads(int(n))


def reduce(method, values, weights):
    mid, p = method_to_id(method)
    v = np.ascontiguousarray(values, dtype=np.float64)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    ws = np.empty(max(v.size, 1), dtype=np.float64)
    return lib().xo_reduce(mid, p, _p(v, _f64p), _p(w, _f64p), _p(ws, _f64p), v.size)


def regrid_csr(method, source, data, indices, indptr, n_target, parallel_rows=False):
    """make_regrid(func)._regrid -- regridder.py:41-67.  source (K, S) -> (K, T) float64."""
    mid, p = method_to_id(method)
    src = np.ascontiguousarray(source, dtype=np.float64)
    assert src.ndim == 2
    K, S = src.shape
    data = np.ascontiguousarray(data, dtype=np.float64)
    indices = np.ascontiguousarray(indices, dtype=np.int64)
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    out = np.empty((K, n_target), dtype=np.float64)
    rc = lib().xo_regrid_csr(
        mid, p, _p(src, _f64p), K, S, _p(data, _f64p), _p(indices, _i64p), _p(indptr, _i64p),
        n_target, _p(out, _f64p), int(parallel_rows),
    )
    assert rc == 0
    return out


def regrid_coo(source, row, col, n_target):
    src = np.ascontiguousarray(source, dtype=np.float64)
    K, S = src.shape
    row = np.ascontiguousarray(row, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    out = np.empty((K, n_target), dtype=np.float64)
    rc = lib().xo_regrid_coo(_p(src, _f64p), K, S, _p(row, _i64p), _p(col, _i64p), row.size, n_target, _p(out, _f64p))
    assert rc == 0
    return out


def to_csr_indptr(row, n):
    row = np.ascontiguousarray(row, dtype=np.int64)
    indptr = np.empty(n + 1, dtype=np.int64)
    rc = lib().xo_to_csr_indptr(_p(row, _i64p), row.size, n, _p(indptr, _i64p))
    assert rc == 0
    return indptr


def area(node_xy, faces):
    xy, f = _xy(node_xy), _faces(faces)
    out = np.empty(f.shape[0], dtype=np.float64)
    lib().xo_area(_p(xy, _f64p), _p(f, _i64p), f.shape[0], f.shape[1], _p(out, _f64p))
    return out


def centroids(node_xy, faces):
    xy, f = _xy(node_xy), _faces(faces)
    out = np.empty((f.shape[0], 2), dtype=np.float64)
    lib().xo_centroids(_p(xy, _f64p), _p(f, _i64p), f.shape[0], f.shape[1], _p(out, _f64p))
    return out


def replace_interpolated_weights(vertices, faces, face_index, weights, node_to_node_map, node_index_threshold):
    v, f = _xy(vertices), _faces(faces)
    fi = np.ascontiguousarray(face_index, dtype=np.int64)
    assert weights.dtype == np.float64 and weights.flags.c_contiguous
    nm = np.ascontiguousarray(node_to_node_map, dtype=np.int64).reshape(-1, 2)
    lib().xo_replace_interpolated_weights(
        _p(v, _f64p), _p(f, _i64p), f.shape[1], _p(fi, _i64p), _p(weights, _f64p), weights.shape[0],
        _p(nm, _i64p), int(node_index_threshold),
    )
    return weights


def clip_area(subject, clipper):
    a, b = _xy(subject), _xy(clipper)
    return lib().xo_clip_area(_p(a, _f64p), a.shape[0], _p(b, _f64p), b.shape[0])


class CellTree2d:
    """Oracle counterpart of numba_celltree.CellTree2d(vertices, faces, fill_value)."""

    def __init__(self, vertices, faces, fill_value=-1):
        self.vertices = _xy(vertices)
        f = _faces(faces)
        self.n_face, self.m = f.shape
        self._h = lib().xo_tree_create(
            _p(self.vertices, _f64p), self.vertices.shape[0], _p(f, _i64p), f.shape[0], f.shape[1], int(fill_value)
        )
        self.n_candidates = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().xo_tree_destroy(self._h)
            self._h = None

    @property
    def faces(self):
        out = np.empty((self.n_face, self.m), dtype=np.int64)
        lib().xo_tree_faces(self._h, _p(out, _i64p))
        return out

    def intersect_faces(self, vertices, faces, fill_value=-1, use_sat=True):
        xy, f = _xy(vertices), _faces(faces)
        nnz = ctypes.c_int64(0)
        ncand = ctypes.c_int64(0)
        rc = lib().xo_intersect_faces_count(
            self._h, _p(xy, _f64p), xy.shape[0], _p(f, _i64p), f.shape[0], f.shape[1], int(fill_value),
            int(use_sat), ctypes.byref(nnz), ctypes.byref(ncand),
        )
        assert rc == 0, rc
        self.n_candidates = ncand.value
        n = nnz.value
        q = np.empty(n, dtype=np.int64)
        s = np.empty(n, dtype=np.int64)
        a = np.empty(n, dtype=np.float64)
        lib().xo_intersect_faces_fill(self._h, _p(q, _i64p), _p(s, _i64p), _p(a, _f64p))
        return q, s, a

    def intersect_faces_bruteforce(self, vertices, faces, fill_value=-1):
        xy, f = _xy(vertices), _faces(faces)
        cap = f.shape[0] * self.n_face
        q = np.empty(cap, dtype=np.int64)
        s = np.empty(cap, dtype=np.int64)
        a = np.empty(cap, dtype=np.float64)
        nnz = ctypes.c_int64(0)
        rc = lib().xo_intersect_faces_bruteforce(
            self._h, _p(xy, _f64p), xy.shape[0], _p(f, _i64p), f.shape[0], f.shape[1], int(fill_value),
            cap, _p(q, _i64p), _p(s, _i64p), _p(a, _f64p), ctypes.byref(nnz),
        )
        assert rc == 0, rc
        n = nnz.value
        return q[:n].copy(), s[:n].copy(), a[:n].copy()

    def intersect_edges(self, edge_coords):
        """numba_celltree CellTree2d.intersect_edges: (n_edge, 2, 2) -> (edge_index, face_index, intersections
        (n, 2, 2)), ordered by (edge, face).  Called from xugrid/regrid/unstructured.py:203-215."""
        xy = np.ascontiguousarray(edge_coords, dtype=np.float64)
        assert xy.ndim == 3 and xy.shape[1:] == (2, 2)
        n = ctypes.c_int64(0)
        rc = lib().xo_intersect_edges_count(self._h, _p(xy, _f64p), xy.shape[0], ctypes.byref(n))
        assert rc == 0, rc
        e = np.empty(n.value, dtype=np.int64)
        f = np.empty(n.value, dtype=np.int64)
        x = np.empty((n.value, 2, 2), dtype=np.float64)
        lib().xo_intersect_edges_fill(self._h, _p(e, _i64p), _p(f, _i64p), _p(x, _f64p))
        return e, f, x

    def default_tolerance(self):
        return lib().xo_default_tolerance(self._h)

    def locate_points(self, points, tolerance=None):
        pts = _xy(points)
        out = np.empty(pts.shape[0], dtype=np.int64)
        tol = -1.0 if tolerance is None else float(tolerance)
        rc = lib().xo_locate_points(self._h, _p(pts, _f64p), pts.shape[0], tol, _p(out, _i64p))
        assert rc == 0
        return out

    def compute_barycentric_weights(self, points, tolerance=None):
        pts = _xy(points)
        fi = np.empty(pts.shape[0], dtype=np.int64)
        w = np.empty((pts.shape[0], self.m), dtype=np.float64)
        tol = -1.0 if tolerance is None else float(tolerance)
        rc = lib().xo_barycentric(self._h, _p(pts, _f64p), pts.shape[0], tol, _p(fi, _i64p), _p(w, _f64p))
        assert rc == 0
        return fi, w
