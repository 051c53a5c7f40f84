"""
Thin object layer over the C ABI: device-resident meshes and CSR weights.

``DeviceMesh`` is the HBM counterpart of ``numba_celltree.CellTree2d(vertices, faces, fill)``
(xugrid/ugrid/ugrid2d.py:908-921); ``DeviceCSR`` is the HBM counterpart of ``MatrixCSR``
(xugrid/core/sparse.py:81-137).  Arrays in, arrays out: float64 / np.intp as in the reference
(xugrid/constants.py:9-10).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import XR_F32, XR_F64, check

IntDType = np.intp
FloatDType = np.float64

# reducer ids of include/xugrid_amd.h
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
    "select": 10,  # CentroidLocatorRegridder: value of the row's last entry (include/xugrid_amd.h XR_SELECT)
}


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def init(device=0):
    """Bind this process to one GPU (one process per GPU)."""
    check(_lib.load().xr_init(int(device)))


def set_stream(handle=None, async_dev=False):
    """Run the engine on the caller's HIP stream (``handle``: the integer stream handle, e.g.
    ``torch.cuda.current_stream().cuda_stream``; 0 is the null stream), or back on its own (``handle=None``).
    ``async_dev``: the ``*_dev`` calls return without waiting for their kernels."""
    if handle is None:
        check(_lib.load().xr_set_stream(None, 0, 0))
    else:
        check(_lib.load().xr_set_stream(ctypes.c_void_p(int(handle)), 1, int(bool(async_dev))))


def set_async(on=True):
    """Asynchronous mode on the engine's own stream (include/xugrid_amd.h: xr_set_async): ``*_dev`` calls,
    ``DeviceMesh.overlap_apply_dev`` and ``invalidate`` return with their kernels in flight; ``dev_sync()`` completes them."""
    check(_lib.load().xr_set_async(1 if on else 0))


def set_option(name, value):
    """A run-time option of the library (include/xugrid_amd.h: xr_set_option; the table is DESIGN.md section 8).  Options start
    from the environment (``XR_<NAME>``, read once); tests and measurement scripts change them here.  -> the previous value."""
    previous = get_option(name)
    check(_lib.load().xr_set_option(name.encode(), int(value)))
    return previous


def get_option(name):
    value = ctypes.c_int64()
    check(_lib.load().xr_get_option(name.encode(), ctypes.byref(value)))
    return int(value.value)


class option:
    """``with engine.option("overlap_fused", 0): ...`` -- an option changed for the block, restored behind it."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.previous = set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.previous)
        return False


# ---- arrays that already live in HBM -----------------------------------------------------------------------------------------
_DEV_TYPESTR = {"<f8": np.float64, "<f4": np.float32, "<i8": np.int64, "<i4": np.int32}


def device_array_info(obj):
    """-> (device pointer, shape, numpy dtype) if ``obj`` is a C-contiguous array in device memory -- a torch tensor on the GPU,
    or anything with ``__cuda_array_interface__`` (cupy, numba, ``DeviceArray``; on ROCm the interface carries HIP pointers) --
    else None.  Device arrays that are not contiguous, or of another dtype than float64 / float32 / int64 / int32, raise."""
    if isinstance(obj, np.ndarray):
        return None
    if (type(obj).__module__ or "").startswith("torch"):
        if not getattr(obj, "is_cuda", False):
            return None
        if not obj.is_contiguous():
            raise ValueError("device arrays must be C-contiguous")
        name = str(obj.dtype).replace("torch.", "")
        table = {"float64": np.float64, "float32": np.float32, "int64": np.int64, "int32": np.int32}
        if name not in table:
            raise TypeError(f"unsupported device dtype {obj.dtype}")
        return int(obj.data_ptr()), tuple(int(n) for n in obj.shape), np.dtype(table[name])
    try:
        cai = getattr(obj, "__cuda_array_interface__", None)
    except Exception:  # noqa: BLE001  (objects that raise for host data)
        cai = None
    if not isinstance(cai, dict):
        return None
    shape = tuple(int(n) for n in cai["shape"])
    typestr = cai["typestr"].replace("=", "<")
    if typestr not in _DEV_TYPESTR:
        raise TypeError(f"unsupported device dtype {cai['typestr']}")
    dtype = np.dtype(_DEV_TYPESTR[typestr])
    strides = cai.get("strides")
    if strides is not None:
        expect, acc = [], dtype.itemsize
        for n in reversed(shape):
            expect.append(acc)
            acc *= max(n, 1)
        if tuple(strides) != tuple(reversed(expect)):
            raise ValueError("device arrays must be C-contiguous")
    return int(cai["data"][0]), shape, dtype


class DeviceArray:
    """An array in the engine's HBM (xr_dev_alloc): what ``Regridder.regrid`` returns for device input that is not a torch
    tensor, and a way to put host arrays there once.  Exposes ``__cuda_array_interface__`` (version 3), so torch / cupy can wrap
    it without a copy; ``download()`` -> numpy."""

    def __init__(self, shape, dtype=np.float64):
        self.shape = tuple(int(n) for n in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        handle = ctypes.c_void_p()
        check(_lib.load().xr_dev_alloc(max(self.nbytes, 1), ctypes.byref(handle)))
        self._h = handle
        self.ptr = int(handle.value or 0)

    @classmethod
    def from_host(cls, array):
        a = np.ascontiguousarray(array)
        out = cls(a.shape, a.dtype)
        if a.nbytes:
            check(_lib.load().xr_dev_upload(out._h, a.ctypes.data_as(ctypes.c_void_p), a.nbytes))
        return out

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 3, "strides": None}

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if int(np.prod(shape, dtype=np.int64)) * self.dtype.itemsize != self.nbytes:
            raise ValueError("cannot reshape a device array to another size")
        view = object.__new__(DeviceArray)
        view.shape, view.dtype, view.nbytes, view.ptr, view._h, view._base = tuple(int(n) for n in shape), self.dtype, self.nbytes, self.ptr, None, self
        return view

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            check(_lib.load().xr_dev_download(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(self.ptr), self.nbytes))
        return out

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.load().xr_dev_free(h)
            except Exception:  # noqa: BLE001
                pass
            self._h = None


def empty_like_device(like, shape, dtype=np.float64):
    """A fresh device array of ``shape`` of the same KIND as ``like``: a torch tensor on ``like``'s device for a torch tensor,
    a ``DeviceArray`` otherwise.  -> (array, device pointer)"""
    if (type(like).__module__ or "").startswith("torch"):
        import torch

        out = torch.empty(tuple(shape), dtype=getattr(torch, np.dtype(dtype).name), device=like.device)
        return out, int(out.data_ptr())
    out = DeviceArray(shape, dtype)
    return out, out.ptr


def sync_producer(obj):
    """Device input handed over by another library was written on ITS stream; the engine reads it on its own.  For torch
    tensors the current torch stream is drained first (a no-op when the engine runs on that very stream, xr_set_stream)."""
    if (type(obj).__module__ or "").startswith("torch"):
        import torch

        torch.cuda.current_stream(obj.device).synchronize()


def _as_xy(vertices):
    xy = np.ascontiguousarray(vertices, dtype=np.float64)
    if xy.ndim != 2 or xy.shape[1] != 2:
        raise ValueError(f"expected an (n, 2) array of coordinates, received shape {xy.shape}")
    return xy


def _as_faces(faces):
    f = np.asarray(faces)
    if f.ndim != 2:
        raise ValueError(f"expected an (n_face, n_max_node) connectivity, received shape {f.shape}")
    if not np.issubdtype(f.dtype, np.integer):
        raise TypeError(f"face_node_connectivity must be integer, received {f.dtype}")
    if f.dtype.itemsize not in (4, 8) or f.dtype.kind != "i":
        f = f.astype(np.int64)
    return np.ascontiguousarray(f)


class DeviceMesh:
    """Face topology resident in HBM (+ lazily built derived data and spatial index)."""

    @classmethod
    def _from_handle(cls, handle):
        """Wrap a mesh that was built on the device (xr_voronoi_mesh)."""
        self = cls.__new__(cls)
        self._h = handle
        n_node, n_face, m = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(_lib.load().xr_mesh_info(handle, ctypes.byref(n_node), ctypes.byref(n_face), ctypes.byref(m)))
        self.n_node, self.n_face, self.n_max_node = n_node.value, n_face.value, m.value
        return self

    @classmethod
    def from_rectilinear(cls, x_vertices, y_vertices):
        """The quad mesh of a rectilinear grid generated on the device from its two 1-D vertex arrays (cell edges
        in the raster's own order): include/xugrid_amd.h, xr_mesh_create_rectilinear."""
        xv = np.ascontiguousarray(x_vertices, dtype=np.float64)
        yv = np.ascontiguousarray(y_vertices, dtype=np.float64)
        if xv.ndim != 1 or yv.ndim != 1 or xv.size < 2 or yv.size < 2:
            raise ValueError("x_vertices and y_vertices must be 1-D arrays of at least two vertices")
        handle = ctypes.c_void_p()
        check(_lib.load().xr_mesh_create_rectilinear(_ptr(xv), xv.size - 1, _ptr(yv), yv.size - 1, ctypes.byref(handle)))
        return cls._from_handle(handle)

    @classmethod
    def from_device(cls, node_xy_ptr, n_node, faces_ptr, faces_itemsize, n_face, n_max_node, fill_value=-1):
        """A mesh from arrays that already live in the engine's HBM (device pointers as integers, e.g.
        ``tensor.data_ptr()``): float64 (n_node, 2) coordinates, int32 / int64 (n_face, n_max_node) connectivity.
        The handle copies both (include/xugrid_amd.h: xr_mesh_create_dev)."""
        handle = ctypes.c_void_p()
        check(
            _lib.load().xr_mesh_create_dev(
                ctypes.c_void_p(int(node_xy_ptr)), int(n_node), ctypes.c_void_p(int(faces_ptr)), int(faces_itemsize),
                int(n_face), int(n_max_node), int(fill_value), ctypes.byref(handle),
            )
        )
        self = cls.__new__(cls)
        self._h = handle
        self.n_node, self.n_face, self.n_max_node = int(n_node), int(n_face), int(n_max_node)
        return self

    def download(self):
        """-> (node_xy float64[n_node, 2], faces int64[n_face, n_max_node]) as uploaded / assembled."""
        xy = np.empty((self.n_node, 2), dtype=np.float64)
        faces = np.empty((self.n_face, self.n_max_node), dtype=np.int64)
        check(_lib.load().xr_mesh_download(self._h, _ptr(xy), _ptr(faces)))
        return xy, faces

    def __init__(self, vertices, faces, fill_value=-1):
        lib = _lib.load()
        xy = _as_xy(vertices)
        f = _as_faces(faces)
        self.n_node = xy.shape[0]
        self.n_face, self.n_max_node = f.shape
        handle = ctypes.c_void_p()
        check(
            lib.xr_mesh_create(
                _ptr(xy), self.n_node, _ptr(f), f.dtype.itemsize, self.n_face, self.n_max_node,
                int(fill_value), ctypes.byref(handle),
            )
        )
        self._h = handle

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.load().xr_mesh_destroy(h)
            except Exception:
                pass
            self._h = None

    def prepare(self):
        check(_lib.load().xr_mesh_prepare(self._h))

    def build_index(self):
        check(_lib.load().xr_mesh_build_index(self._h))

    def invalidate(self):
        check(_lib.load().xr_mesh_invalidate(self._h))

    def device_bytes(self):
        """HBM currently held by the handle (raw + prepared + query order + tree index)."""
        n = ctypes.c_int64(0)
        check(_lib.load().xr_mesh_device_bytes(self._h, ctypes.byref(n)))
        return n.value

    def area(self):
        out = np.empty(self.n_face, dtype=np.float64)
        check(_lib.load().xr_mesh_area(self._h, _ptr(out)))
        return out

    def centroids(self):
        out = np.empty((self.n_face, 2), dtype=np.float64)
        check(_lib.load().xr_mesh_centroids(self._h, _ptr(out)))
        return out

    def faces_ccw(self):
        out = np.empty((self.n_face, self.n_max_node), dtype=np.int64)
        check(_lib.load().xr_mesh_faces(self._h, _ptr(out)))
        return out

    def overlap(self, query: "DeviceMesh", relative=False) -> "DeviceCSR":
        """All (query face, self face) pairs with positive intersection area, as CSR rows=query."""
        handle = ctypes.c_void_p()
        check(_lib.load().xr_overlap(self._h, query._h, int(bool(relative)), ctypes.byref(handle)))
        return DeviceCSR(handle)

    def overlap_apply_dev(self, query: "DeviceMesh", source_ptr, source_dtype, K, out_ptr, method_id=0, percentile=0.0,
                          relative=False) -> "DeviceCSR":
        """``overlap(query)`` and the apply of ``K`` variables in one call (device pointers in and out): the weights'
        first use rides on their construction (include/xugrid_amd.h: xr_overlap_apply_dev).  -> the matrix."""
        handle = ctypes.c_void_p()
        check(_lib.load().xr_overlap_apply_dev(self._h, query._h, int(bool(relative)), int(method_id), float(percentile),
                                               ctypes.c_void_p(int(source_ptr)), int(source_dtype), int(K),
                                               ctypes.c_void_p(int(out_ptr)), ctypes.byref(handle)))
        return DeviceCSR(handle)

    def overlap_partial_dev(self, query: "DeviceMesh", source_ptr, source_dtype, K, out_ptr, method_id, rows_layout,
                            relative=False) -> "DeviceCSR":
        """``overlap(query)`` and the per-target partial STATE of ``K`` variables in one call (a rank of the sharded
        regridder; include/xugrid_amd.h: xr_overlap_partial_dev).  -> the matrix."""
        handle = ctypes.c_void_p()
        check(_lib.load().xr_overlap_partial_dev(self._h, query._h, int(bool(relative)), int(method_id),
                                                 ctypes.c_void_p(int(source_ptr)), int(source_dtype), int(K),
                                                 ctypes.c_void_p(int(out_ptr)), int(bool(rows_layout)), ctypes.byref(handle)))
        return DeviceCSR(handle)

    def last_candidates(self):
        n = ctypes.c_int64(0)
        check(_lib.load().xr_overlap_stats(self._h, ctypes.byref(n)))
        return n.value

    def locate_points(self, points, tolerance=None):
        pts = _as_xy(points)
        out = np.empty(pts.shape[0], dtype=np.int64)
        tol = -1.0 if tolerance is None else float(tolerance)
        if tolerance is not None and tol < 0:
            raise ValueError("tolerance must be non-negative")
        check(_lib.load().xr_locate_points(self._h, _ptr(pts), pts.shape[0], tol, _ptr(out)))
        return out.astype(IntDType, copy=False)

    def locate_raster(self, x, y, tolerance=None):
        """locate_points on the nodes (x[i], y[j]) of a raster -> index int (len(y), len(x)), -1 outside."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        if x.ndim != 1 or y.ndim != 1:
            raise ValueError("x and y must be 1-D")
        tol = -1.0 if tolerance is None else float(tolerance)
        if tolerance is not None and tol < 0:
            raise ValueError("tolerance must be non-negative")
        out = np.empty((y.size, x.size), dtype=np.int64)
        check(_lib.load().xr_locate_raster(self._h, _ptr(x), x.size, _ptr(y), y.size, tol, _ptr(out)))
        return out.astype(IntDType, copy=False)

    def compute_barycentric_weights(self, points, tolerance=None):
        pts = _as_xy(points)
        face = np.empty(pts.shape[0], dtype=np.int64)
        w = np.empty((pts.shape[0], self.n_max_node), dtype=np.float64)
        tol = -1.0 if tolerance is None else float(tolerance)
        if tolerance is not None and tol < 0:
            raise ValueError("tolerance must be non-negative")
        check(_lib.load().xr_barycentric(self._h, _ptr(pts), pts.shape[0], tol, _ptr(face), _ptr(w)))
        return face.astype(IntDType, copy=False), w


class DeviceVoronoi:
    """Device part of the centroidal Voronoi pre-step of a mesh (see include/xugrid_amd.h, xr_voronoi_*)."""

    def __init__(self, mesh: DeviceMesh):
        handle = ctypes.c_void_p()
        check(_lib.load().xr_voronoi_create(mesh._h, ctypes.byref(handle)))
        self._h = handle
        self._mesh = mesh  # keep the source alive
        vals = [ctypes.c_int64() for _ in range(5)]
        check(_lib.load().xr_voronoi_info(handle, *[ctypes.byref(v) for v in vals]))
        self.n_node, self.nnz, self.n_exterior_edge, self.n_interior_cell, self.max_interior_degree = (v.value for v in vals)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.load().xr_voronoi_destroy(h)
            except Exception:
                pass
            self._h = None

    def download(self):
        """-> (indptr, indices) of node_face_connectivity, exterior edge_nodes (n, 2), edge_face (n,), centroids."""
        indptr = np.empty(self.n_node + 1, dtype=np.int64)
        indices = np.empty(self.nnz, dtype=np.int64)
        edge_nodes = np.empty((self.n_exterior_edge, 2), dtype=np.int64)
        edge_face = np.empty(self.n_exterior_edge, dtype=np.int64)
        centroids = np.empty((self._mesh.n_face, 2), dtype=np.float64)
        check(
            _lib.load().xr_voronoi_download(
                self._h, _ptr(indptr), _ptr(indices), _ptr(edge_nodes), _ptr(edge_face), _ptr(centroids)
            )
        )
        return indptr, indices, edge_nodes, edge_face, centroids

    def download_boundary(self):
        """What the O(boundary) host part of the Voronoi step reads, and nothing else: ``(nodes, row_ptr, faces,
        face_xy, edge_nodes, edge_face, edge_face_xy)`` -- the boundary nodes (ascending), their rows of
        node_face_connectivity as a small CSR with the centroid of every listed face, the exterior edges and the
        centroid of each edge's face."""
        nb, ne = ctypes.c_int64(), ctypes.c_int64()
        check(_lib.load().xr_voronoi_boundary_info(self._h, ctypes.byref(nb), ctypes.byref(ne)))
        nodes = np.empty(nb.value, dtype=np.int64)
        row_ptr = np.empty(nb.value + 1, dtype=np.int64)
        faces = np.empty(ne.value, dtype=np.int64)
        face_xy = np.empty((ne.value, 2), dtype=np.float64)
        edge_nodes = np.empty((self.n_exterior_edge, 2), dtype=np.int64)
        edge_face = np.empty(self.n_exterior_edge, dtype=np.int64)
        edge_face_xy = np.empty((self.n_exterior_edge, 2), dtype=np.float64)
        check(
            _lib.load().xr_voronoi_boundary(
                self._h, _ptr(nodes), _ptr(row_ptr), _ptr(faces), _ptr(face_xy), _ptr(edge_nodes), _ptr(edge_face),
                _ptr(edge_face_xy),
            )
        )
        return nodes, row_ptr, faces, face_xy, edge_nodes, edge_face, edge_face_xy

    def boundary_cells(self):
        """The cells of the boundary nodes as the library computes them (native O(boundary) host part):
        ``(extra_xy (n_extra, 2), cells (n_cell, n_max) int64, -1 padded)``."""
        ne, nc, nm = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(_lib.load().xr_voronoi_boundary_cells_info(self._h, ctypes.byref(ne), ctypes.byref(nc), ctypes.byref(nm)))
        extra = np.empty((ne.value, 2), dtype=np.float64)
        cells = np.empty((nc.value, nm.value), dtype=np.int64)
        check(_lib.load().xr_voronoi_boundary_cells(self._h, _ptr(extra), _ptr(cells)))
        return extra, cells

    def assemble_auto(self):
        """-> (DeviceMesh of the tessellation, tail_face_index, interpolation_map): boundary cells by the library,
        assembly on the device (xr_voronoi_mesh_auto)."""
        handle = ctypes.c_void_p()
        nt, nm = ctypes.c_int64(), ctypes.c_int64()
        check(_lib.load().xr_voronoi_mesh_auto(self._h, ctypes.byref(handle), ctypes.byref(nt), ctypes.byref(nm)))
        mesh = DeviceMesh._from_handle(handle)
        tail = np.empty(nt.value, dtype=np.int64)
        imap = np.empty((nm.value, 2), dtype=np.int64)
        check(_lib.load().xr_voronoi_tail(self._h, _ptr(tail), _ptr(imap)))
        return mesh, tail, imap

    def assemble(self, extra_xy, boundary_cells) -> DeviceMesh:
        extra_xy = np.ascontiguousarray(extra_xy, dtype=np.float64).reshape(-1, 2)
        cells = np.ascontiguousarray(boundary_cells, dtype=np.int64)
        if cells.ndim != 2:
            raise ValueError("boundary_cells must be 2-D")
        handle = ctypes.c_void_p()
        check(
            _lib.load().xr_voronoi_mesh(
                self._h, _ptr(extra_xy), extra_xy.shape[0], _ptr(cells), cells.shape[0], cells.shape[1], ctypes.byref(handle)
            )
        )
        return DeviceMesh._from_handle(handle)


def morton_row_keys(xy, faces_per_tile=144):
    """Coarse Morton keys of points (row locality hint): tiles holding about ``faces_per_tile`` points each.
    -> (keys int64[n], key_range)"""
    xy = np.asarray(xy, dtype=np.float64)
    n = xy.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64), 1
    lo = np.nanmin(xy, axis=0)
    span = float(np.nanmax(np.nanmax(xy, axis=0) - lo))
    if not span > 0:
        return np.zeros(n, dtype=np.int64), 1
    bits = int(np.clip(np.floor(0.5 * np.log2(max(n / faces_per_tile, 1.0))), 0, 10))
    side = 1 << bits
    cell = np.clip(np.nan_to_num((xy - lo) * (side / (span * (1 + 1e-9)))).astype(np.int64), 0, side - 1)

    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v

    return spread(cell[:, 0]) | (spread(cell[:, 1]) << 1), 1 << (2 * bits)


def edge_length_csr(tree: DeviceMesh, edge_node_coordinates) -> "DeviceCSR":
    """NetworkGridder weights: (n_edge, 2, 2) end points -> CSR rows = faces of ``tree``, columns = edges, data =
    length of the edge inside the face (include/xugrid_amd.h: xr_edge_length_csr)."""
    handle = ctypes.c_void_p()
    info = device_array_info(edge_node_coordinates)
    if info is not None:  # the end points already live in HBM (xr_edge_length_csr_dev): no upload
        ptr, shape, dtype = info
        if len(shape) != 3 or tuple(shape[1:]) != (2, 2) or dtype != np.dtype(np.float64):
            raise ValueError("edge_node_coordinates must be a float64 (n_edge, 2, 2) array")
        sync_producer(edge_node_coordinates)
        check(_lib.load().xr_edge_length_csr_dev(tree._h, ctypes.c_void_p(ptr), shape[0], ctypes.byref(handle)))
        return DeviceCSR(handle)
    xy = np.ascontiguousarray(edge_node_coordinates, dtype=np.float64)
    if xy.ndim != 3 or xy.shape[1:] != (2, 2):
        raise ValueError("edge_node_coordinates must have shape (n_edge, 2, 2)")
    check(_lib.load().xr_edge_length_csr(tree._h, _ptr(xy), xy.shape[0], ctypes.byref(handle)))
    return DeviceCSR(handle)


def edge_pieces(tree: DeviceMesh, csr: "DeviceCSR", edge_node_coordinates):
    """End points (nnz, 2, 2) of the pieces behind the entries of an ``edge_length_csr`` matrix, in entry order."""
    xy = np.ascontiguousarray(edge_node_coordinates, dtype=np.float64)
    out = np.empty((csr.nnz, 2, 2), dtype=np.float64)
    check(_lib.load().xr_edge_pieces(tree._h, csr._h, _ptr(xy), xy.shape[0], _ptr(out)))
    return out


def locate_csr(tree: DeviceMesh, query: DeviceMesh = None, points=None, tolerance=None) -> "DeviceCSR":
    """locate_centroids + MatrixCOO.from_triplet on the device: one (face, 1.0) entry per located point."""
    tol = -1.0 if tolerance is None else float(tolerance)
    if tolerance is not None and tol < 0:
        raise ValueError("tolerance must be non-negative")
    if (query is None) == (points is None):
        raise ValueError("give either a query mesh or points")
    if points is not None:
        pts = _as_xy(points)
        p_arg, n, q_arg = _ptr(pts), pts.shape[0], None
    else:
        p_arg, n, q_arg = None, query.n_face, query._h
    handle = ctypes.c_void_p()
    check(_lib.load().xr_locate_csr(tree._h, q_arg, p_arg, n, tol, ctypes.byref(handle)))
    return DeviceCSR(handle)


def replace_interpolated_weights(vertices, faces, face_index, weights, node_to_node_map, node_index_threshold):
    """xugrid/regrid/unstructured.py:17-57 on the device; ``weights`` (float64, C-contiguous) is updated in place."""
    v = _as_xy(vertices)
    f = np.ascontiguousarray(faces, dtype=np.int64)
    fi = np.ascontiguousarray(face_index, dtype=np.int64)
    nm = np.ascontiguousarray(node_to_node_map, dtype=np.int64).reshape(-1, 2)
    if weights.dtype != np.float64 or not weights.flags.c_contiguous or weights.ndim != 2:
        raise ValueError("weights must be a C-contiguous float64 (n, m) array (it is updated in place)")
    if int(node_index_threshold) != v.shape[0] - nm.shape[0]:
        raise ValueError("node_index_threshold must be len(vertices) - len(node_to_node_map)")
    if weights.shape != (fi.size, f.shape[1]):
        raise ValueError("weights must have one row per point and one column per face slot")
    check(_lib.load().xr_replace_interpolated_weights(
        _ptr(v), v.shape[0], _ptr(f), f.shape[0], f.shape[1], _ptr(fi), _ptr(weights), fi.size, _ptr(nm), nm.shape[0]))
    return weights


class DevicePoints:
    """Query points of a barycentric construction and their "inside the source grid" flags, in HBM
    (include/xugrid_amd.h: xr_locate_flags_begin).  The source-side kernels are deferred: the engine enqueues them when the
    Voronoi pre-step or the construction that consumes the handle next has the device to spare.  The handle keeps BOTH
    meshes alive until then."""

    def __init__(self, source: DeviceMesh, query: DeviceMesh = None, points=None):
        if (query is None) == (points is None):
            raise ValueError("give either a query mesh or points")
        handle = ctypes.c_void_p()
        if points is not None:
            pts = _as_xy(points)
            check(_lib.load().xr_locate_flags_begin(source._h, None, _ptr(pts), pts.shape[0], ctypes.byref(handle)))
            self.n = pts.shape[0]
        else:
            check(_lib.load().xr_locate_flags_begin(source._h, query._h, None, 0, ctypes.byref(handle)))
            self.n = query.n_face
        self._h = handle
        self._source = source  # (the deferred kernels read both meshes: they live as long as the handle)
        self._query = query

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.load().xr_points_destroy(h)
            except Exception:
                pass
            self._h = None


def barycentric_csr(voronoi: DeviceMesh, source: DeviceMesh, vertex_face, node_to_node_map, query: DeviceMesh = None,
                    points=None, tolerance=None, n_identity=0, reference_order=True, prepared: "DevicePoints" = None) -> "DeviceCSR":
    """UnstructuredGrid2d.barycentric after the Voronoi pre-step, on the device (see include/xugrid_amd.h).
    ``n_identity`` > 0: ``vertex_face`` holds only the entries of the vertices ``>= n_identity`` (the first
    ``n_identity`` vertices are the source face centroids, in face order).  ``reference_order`` (default): pair the
    weight slots with the caller's vertex order of every cell, as the reference does (unstructured.py:175,193);
    False = the tree's own counter-clockwise order (opt-in, not the reference's result)."""
    vertex_face = np.ascontiguousarray(vertex_face, dtype=np.int64)
    if vertex_face.shape != (voronoi.n_node - n_identity,):
        raise ValueError("vertex_face must have one entry per Voronoi vertex")
    if node_to_node_map is None:
        n2n = np.zeros((0, 2), dtype=np.int64)
    else:
        n2n = np.ascontiguousarray(node_to_node_map, dtype=np.int64).reshape(-1, 2)
    tol = -1.0 if tolerance is None else float(tolerance)
    if tolerance is not None and tol < 0:
        raise ValueError("tolerance must be non-negative")
    handle = ctypes.c_void_p()
    if prepared is not None:
        # ``prepared``: a DevicePoints started earlier for the same source and query (its kernels ran beside whatever the
        # host did in between, e.g. the boundary cells of the Voronoi pre-step)
        if query is not None or points is not None:
            raise ValueError("prepared points replace the query mesh / points")
        check(
            _lib.load().xr_barycentric_csr_points(
                voronoi._h, source._h, prepared._h, tol, int(n_identity), _ptr(vertex_face), _ptr(n2n), n2n.shape[0],
                0 if reference_order else 1, ctypes.byref(handle),
            )
        )
        return DeviceCSR(handle)
    if (query is None) == (points is None):
        raise ValueError("give either a query mesh or points")
    if points is not None:
        pts = _as_xy(points)
        p_arg, n, q_arg = _ptr(pts), pts.shape[0], None
    else:
        p_arg, n, q_arg = None, query.n_face, query._h
    if n_identity or not reference_order:
        check(
            _lib.load().xr_barycentric_csr_tail(
                voronoi._h, source._h, q_arg, p_arg, n, tol, int(n_identity), _ptr(vertex_face), _ptr(n2n), n2n.shape[0],
                0 if reference_order else 1, ctypes.byref(handle),
            )
        )
    else:
        check(
            _lib.load().xr_barycentric_csr(
                voronoi._h, source._h, q_arg, p_arg, n, tol, _ptr(vertex_face), _ptr(n2n), n2n.shape[0],
                ctypes.byref(handle),
            )
        )
    return DeviceCSR(handle)


def _source_2d(source):
    """(K, S) C-contiguous float64/float32 view of the source block (regridder.py:152-163)."""
    a = np.asarray(source)
    if a.ndim != 2:
        raise ValueError(f"source must be 2-D (n_extra, size), received shape {a.shape}")
    if a.dtype == np.float32:
        return np.ascontiguousarray(a), XR_F32
    if a.dtype != np.float64:
        # ints / bools / float16: the reference copies into a float64 workspace (regridder.py:49,57-58)
        a = a.astype(np.float64)
    return np.ascontiguousarray(a), XR_F64


def _axis_arrays(axis_y, axis_x):
    arrays = []
    for indptr, source, weight in (axis_y, axis_x):
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        source = np.ascontiguousarray(source, dtype=np.int64)
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        if indptr.ndim != 1 or indptr.size < 1 or source.shape != weight.shape or source.ndim != 1:
            raise ValueError("inconsistent axis arrays")
        if indptr[-1] != source.size:
            raise ValueError("axis indptr does not span its entries")
        arrays.append((indptr, source, weight))
    return arrays


class DeviceCSR:
    """MatrixCSR resident in HBM: rows = target faces, columns = source faces."""

    def __init__(self, handle, owner=None):
        self._h = handle
        self._owner = owner  # a DeviceOuter that owns the handle (borrowed view): keep it alive, never destroy
        n, m, nnz = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(_lib.load().xr_csr_info(handle, ctypes.byref(n), ctypes.byref(m), ctypes.byref(nnz)))
        self.n, self.m, self.nnz = n.value, m.value, nnz.value

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and getattr(self, "_owner", None) is None:
            try:
                _lib.load().xr_csr_destroy(h)
            except Exception:
                pass
            self._h = None

    @classmethod
    def from_arrays(cls, data, indices, indptr, n, m):
        data = np.ascontiguousarray(data, dtype=np.float64)
        indices = np.ascontiguousarray(indices, dtype=np.int64)
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        if indptr.size != n + 1 or indices.size != data.size:
            raise ValueError("inconsistent CSR arrays")
        handle = ctypes.c_void_p()
        check(
            _lib.load().xr_csr_upload(
                _ptr(data), _ptr(indices), _ptr(indptr), int(n), int(m), data.size, ctypes.byref(handle)
            )
        )
        return cls(handle)

    @classmethod
    def from_triplet(cls, row, col, data, n, m):
        row = np.ascontiguousarray(row, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int64)
        data = np.ascontiguousarray(data, dtype=np.float64)
        handle = ctypes.c_void_p()
        check(
            _lib.load().xr_csr_from_triplet(
                _ptr(row), _ptr(col), _ptr(data), data.size, int(n), int(m), ctypes.byref(handle)
            )
        )
        return cls(handle)

    @classmethod
    def from_outer(cls, axis_y, n_source_y, axis_x, n_source_x):
        """
        CSR of the outer product of two per-axis sparse matrices, each ``(indptr, source, weight)`` with
        the source indices ascending within a row (StructuredGrid2d.broadcast_sorted, structured.py:503-531).
        """
        (ipy, sy, wy), (ipx, sx, wx) = _axis_arrays(axis_y, axis_x)
        handle = ctypes.c_void_p()
        check(
            _lib.load().xr_csr_from_outer(
                _ptr(ipy), _ptr(sy), _ptr(wy), ipy.size - 1, int(n_source_y),
                _ptr(ipx), _ptr(sx), _ptr(wx), ipx.size - 1, int(n_source_x), ctypes.byref(handle),
            )
        )
        return cls(handle)

    def set_row_keys(self, keys, key_range):
        """Locality hint for uploaded weights (see include/xugrid_amd.h: xr_csr_set_row_keys)."""
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        if keys.shape != (self.n,):
            raise ValueError("one key per row expected")
        check(_lib.load().xr_csr_set_row_keys(self._h, _ptr(keys), int(key_range)))

    def set_col_keys(self, keys, key_range):
        """Renumber the columns (source cells) by a spatial key (see include/xugrid_amd.h: xr_csr_set_col_keys)."""
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        if keys.shape != (self.m,):
            raise ValueError("one key per column expected")
        check(_lib.load().xr_csr_set_col_keys(self._h, _ptr(keys), int(key_range)))

    def col_order(self):
        """stored column j holds the caller's column ``col_order()[j]``"""
        out = np.empty(self.m, dtype=np.int64)
        check(_lib.load().xr_csr_col_order(self._h, _ptr(out)))
        return out

    def expect_permuted(self, permuted=True):
        """The source blocks of the following applies are already in the stored column order
        (``source[:, col_order()]``)."""
        check(_lib.load().xr_csr_expect_permuted(self._h, 1 if permuted else 0))

    def engine_order(self, source_xy, target_xy, K=256, row_tile=4, col_tile=8):
        """Put BOTH sides of the many-variable apply in the engine's own order: rows (target cells) and columns (source
        cells) renumbered along Morton curves of the given centroids, source blocks expected in the stored column order and
        results delivered in the stored row order.  -> (col_order, row_order): feed ``source[:, col_order]``, read
        ``out[:, r]`` as the caller's row ``row_order[r]``.  For pipelines that keep their (K, S) / (K, T) blocks on the
        device across many applies the caller's numbering is then paid once, not per apply (1M x 1M benchmark matrix,
        K = 256: 1.83 -> 1.13 ms = 46 % of HBM)."""
        rk, rr = morton_row_keys(target_xy, faces_per_tile=row_tile)
        self.set_row_keys(rk, rr)
        ck, cr = morton_row_keys(source_xy, faces_per_tile=col_tile)
        self.set_col_keys(ck, cr)
        self.expect_permuted(True)
        self.output_stored_order(True)
        return self.col_order(), self.row_order()

    def output_stored_order(self, stored=True):
        """The following applies write their rows in the STORED order (``out[:, r]`` = caller's row ``row_order()[r]``)."""
        check(_lib.load().xr_csr_output_stored_order(self._h, 1 if stored else 0))

    def row_order(self, K=None):
        """stored row r holds the caller's row ``row_order()[r]``.  The pending regrouping of the rows into tiles (the
        many-variable apply) is settled by this call whatever the number of variables of the coming applies (``K`` is
        accepted and ignored), so the permutation returned here stays valid."""
        out = np.empty(self.n, dtype=np.int64)
        check(_lib.load().xr_csr_row_order(self._h, 0, _ptr(out)))
        return out

    def download(self):
        """-> (data float64[nnz], indices intp[nnz], indptr intp[n+1])"""
        data = np.empty(self.nnz, dtype=np.float64)
        indices = np.empty(self.nnz, dtype=np.int64)
        indptr = np.empty(self.n + 1, dtype=np.int64)
        check(_lib.load().xr_csr_download(self._h, _ptr(data), _ptr(indices), _ptr(indptr)))
        return data, indices.astype(IntDType, copy=False), indptr.astype(IntDType, copy=False)

    def apply(self, source, method_id=0, percentile=0.0, out=None):
        """make_regrid(f)._regrid(source, A, size): (K, S) -> float64 (K, T).  ``out``: optional preallocated
        C-contiguous float64 (K, T) array to reuse (for large K the first touch and the release of a fresh result
        array cost more than the transfer itself)."""
        src, dtype = _source_2d(source)
        if src.shape[1] != self.m:
            raise ValueError(f"source has {src.shape[1]} cells, weights expect {self.m}")
        K = src.shape[0]
        if out is None:
            out = np.empty((K, self.n), dtype=np.float64)
        elif out.shape != (K, self.n) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous float64 array of shape {(K, self.n)}")
        check(_lib.load().xr_apply_csr(self._h, int(method_id), float(percentile), _ptr(src), dtype, K, _ptr(out)))
        return out

    def apply_dev(self, source_ptr, dtype, K, out_ptr, method_id=0, percentile=0.0):
        """Device-pointer variant (e.g. torch tensor .data_ptr()); no host transfer."""
        check(
            _lib.load().xr_apply_csr_dev(
                self._h, int(method_id), float(percentile), ctypes.c_void_p(source_ptr), int(dtype), int(K),
                ctypes.c_void_p(out_ptr),
            )
        )

    def partial_dev(self, source_ptr, dtype, K, out_ptr, method_id, rows_layout):
        """partial reducer state over this matrix' columns (multi-GPU split): planes [C, K, n] or rows [n, C * K]"""
        check(
            _lib.load().xr_apply_partial_dev(
                self._h, int(method_id), ctypes.c_void_p(source_ptr), int(dtype), int(K), ctypes.c_void_p(out_ptr),
                1 if rows_layout else 0,
            )
        )


class DeviceOuter:
    """Separable (rectilinear x rectilinear) weights kept as their two per-axis factors in HBM
    (include/xugrid_amd.h: xr_outer_create).  Same interface as DeviceCSR; the product matrix is only
    materialised for ``download`` / ``csr`` and for the mode / percentile reducers."""

    def __init__(self, axis_y, n_source_y, axis_x, n_source_x):
        (ipy, sy, wy), (ipx, sx, wx) = _axis_arrays(axis_y, axis_x)
        handle = ctypes.c_void_p()
        check(
            _lib.load().xr_outer_create(
                _ptr(ipy), _ptr(sy), _ptr(wy), ipy.size - 1, int(n_source_y),
                _ptr(ipx), _ptr(sx), _ptr(wx), ipx.size - 1, int(n_source_x), ctypes.byref(handle),
            )
        )
        self._h = handle
        n, m, nnz = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(_lib.load().xr_outer_info(handle, ctypes.byref(n), ctypes.byref(m), ctypes.byref(nnz)))
        self.n, self.m, self.nnz = n.value, m.value, nnz.value

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.load().xr_outer_destroy(h)
            except Exception:
                pass
            self._h = None

    def csr(self) -> "DeviceCSR":
        """The materialised product (built once, owned by this object)."""
        handle = ctypes.c_void_p()
        check(_lib.load().xr_outer_csr(self._h, ctypes.byref(handle)))
        return DeviceCSR(handle, owner=self)

    def download(self):
        return self.csr().download()

    def apply(self, source, method_id=0, percentile=0.0, out=None):
        src, dtype = _source_2d(source)
        if src.shape[1] != self.m:
            raise ValueError(f"source has {src.shape[1]} cells, weights expect {self.m}")
        K = src.shape[0]
        if out is None:
            out = np.empty((K, self.n), dtype=np.float64)
        elif out.shape != (K, self.n) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous float64 array of shape {(K, self.n)}")
        check(_lib.load().xr_apply_outer(self._h, int(method_id), float(percentile), _ptr(src), dtype, K, _ptr(out)))
        return out

    def apply_dev(self, source_ptr, dtype, K, out_ptr, method_id=0, percentile=0.0):
        check(
            _lib.load().xr_apply_outer_dev(
                self._h, int(method_id), float(percentile), ctypes.c_void_p(source_ptr), int(dtype), int(K),
                ctypes.c_void_p(out_ptr),
            )
        )


SHARD_MODES = {"hash": 0, "morton": 1, "balanced": 2}


def shard_plan_dev(src_xy_ptr, src_faces_ptr, n_src_face, src_m, tgt_xy_ptr, tgt_faces_ptr, n_tgt_face, tgt_m, world, rank, mode,
                   local_faces_ptr, local_targets_ptr, owner_ptr=0):
    """The partition rule of a rank evaluated on the device (include/xugrid_amd.h: xr_shard_plan_dev): device pointers of the
    replicated raw meshes in, ascending global id lists out.  -> (n_local_faces, n_local_targets)."""
    n_f, n_t = ctypes.c_int64(0), ctypes.c_int64(0)
    check(_lib.load().xr_shard_plan_dev(
        ctypes.c_void_p(int(src_xy_ptr)), ctypes.c_void_p(int(src_faces_ptr)), int(n_src_face), int(src_m),
        ctypes.c_void_p(int(tgt_xy_ptr)), ctypes.c_void_p(int(tgt_faces_ptr)), int(n_tgt_face), int(tgt_m), int(world), int(rank),
        SHARD_MODES[mode], ctypes.c_void_p(int(local_faces_ptr)), ctypes.byref(n_f), ctypes.c_void_p(int(local_targets_ptr)),
        ctypes.byref(n_t), ctypes.c_void_p(int(owner_ptr)) if owner_ptr else None))
    return n_f.value, n_t.value


def partial_components(method_id):
    """number of partial-state components of a shard-decomposable reducer (0: the reducer needs whole rows)"""
    return int(_lib.load().xr_partial_components(int(method_id)))


def partial_combine_is_max(method_id):
    return bool(_lib.load().xr_partial_combine_is_max(int(method_id)))


def partial_fill_identity_dev(method_id, planes_ptr, K, n):
    check(_lib.load().xr_partial_fill_identity_dev(int(method_id), ctypes.c_void_p(planes_ptr), int(K), int(n)))


def finalize_partial_dev(method_id, planes_ptr, K, n, out_ptr):
    check(_lib.load().xr_finalize_partial_dev(int(method_id), ctypes.c_void_p(planes_ptr), int(K), int(n),
                                              ctypes.c_void_p(out_ptr)))


def reduce_partial_rows_dev(method_id, rows_ptr, indptr_ptr, order_ptr, n_targets, K, out_ptr):
    check(_lib.load().xr_reduce_partial_rows_dev(int(method_id), ctypes.c_void_p(rows_ptr), ctypes.c_void_p(indptr_ptr),
                                                 ctypes.c_void_p(order_ptr), int(n_targets), int(K),
                                                 ctypes.c_void_p(out_ptr)))


def apply_coo(row, col, n_target, source):
    """CentroidLocatorRegridder._regrid (regridder.py:400-409) for externally supplied, unsorted COO weights:
    ``out[k, row[i]] = source[k, col[i]]`` in entry order, i.e. the LAST entry of a target wins.  A scatter with one
    thread per entry would race on repeated targets, so the triplets are stably sorted by row and applied as CSR rows
    with the ``select`` reducer (the value of the row's last entry) -- deterministic, and any K."""
    src, _ = _source_2d(source)
    row = np.ascontiguousarray(row, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    if row.shape != col.shape or row.ndim != 1:
        raise ValueError("row and col must be 1-D arrays of equal length")
    if row.size and (row.min() < 0 or row.max() >= n_target or col.min() < 0 or col.max() >= src.shape[1]):
        raise ValueError("COO entry out of range")
    order = np.argsort(row, kind="stable")
    csr = DeviceCSR.from_triplet(row[order], col[order], np.ones(row.size), int(n_target), src.shape[1])
    return csr.apply(src, METHOD_IDS["select"], 0.0)


class KernelTimer:
    """In-library hipEvent timing of every kernel launch on the engine stream."""

    def __enter__(self):
        lib = _lib.load()
        check(lib.xr_prof_enable(1))
        check(lib.xr_prof_reset())
        return self

    def __exit__(self, *exc):
        self.records = kernel_times()
        check(_lib.load().xr_prof_enable(0))
        return False


def kernel_times():
    lib = _lib.load()
    n = ctypes.c_int(0)
    check(lib.xr_prof_count(ctypes.byref(n)))
    out = {}
    for i in range(n.value):
        name = ctypes.create_string_buffer(128)
        launches = ctypes.c_int64(0)
        ms = ctypes.c_double(0.0)
        check(lib.xr_prof_get(i, name, 128, ctypes.byref(launches), ctypes.byref(ms)))
        out[name.value.decode()] = (launches.value, ms.value)
    return out


def prof_enable(on=True):
    check(_lib.load().xr_prof_enable(int(bool(on))))


def prof_reset():
    check(_lib.load().xr_prof_reset())


def dev_sync():
    check(_lib.load().xr_dev_sync())
