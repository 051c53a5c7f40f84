"""
Minimal ``Ugrid2d``: exactly the slice of xugrid/ugrid/ugrid2d.py the regridding hot path touches
(SURVEY.md 2 row 11): the constructor's connectivity handling (:72-115), ``node_coordinates``
(ugridbase.py:576-579), ``area`` (:575-584), ``centroids`` (:544-559), ``celltree`` (:908-921),
``locate_points`` (ugridbase.py:1305-1323), ``compute_barycentric_weights`` (:1054-1078) and
``from_structured_bounds`` (:1894-1912, :1973-2034).  Everything else of the 2.2 kLoC class
(IO, plotting, selection, partitioning ...) is out of scope.
"""
import numpy as np

import ctypes

from . import _lib, connectivity
from .celltree import CellTree2d
from .engine import FloatDType, IntDType

FILL_VALUE = -1


def _node_table(node_x, node_y):
    """(n, 2) C-contiguous float64 table of the node coordinates: the layout the device wants, written once -- by the
    library's host threads when the inputs are float64 (any stride, e.g. the two columns of an (n, 2) array)."""
    x, y = np.asarray(node_x), np.asarray(node_y)
    if x.shape != y.shape or x.ndim != 1:
        raise ValueError("node_x and node_y must be 1-D arrays of equal length")
    n = x.size
    if n >= 65536 and x.dtype == np.float64 and y.dtype == np.float64 and x.strides[0] % 8 == 0 and y.strides[0] % 8 == 0:
        out = np.empty((n, 2), dtype=np.float64)
        _lib.check(_lib.load().xr_host_interleave2(ctypes.c_void_p(x.ctypes.data), x.strides[0] // 8, ctypes.c_void_p(y.ctypes.data),
                                                   y.strides[0] // 8, n, ctypes.c_void_p(out.ctypes.data)))
        return out
    return np.column_stack([np.asarray(x, dtype=FloatDType), np.asarray(y, dtype=FloatDType)])


def _copy_connectivity(faces):
    """face_node_connectivity.copy() as IntDType (ugrid2d.py:94-96): a private copy the constructor may rewrite."""
    if faces.dtype == IntDType and faces.flags.c_contiguous and faces.nbytes >= (1 << 20):
        out = np.empty(faces.shape, dtype=IntDType)
        _lib.check(_lib.load().xr_host_copy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(faces.ctypes.data), faces.nbytes))
        return out
    return faces.astype(IntDType, copy=True)


class Ugrid2d:
    def __init__(self, node_x, node_y, fill_value, face_node_connectivity, name="mesh2d", start_index=0):
        # (n, 2) node table, contiguous: what node_coordinates returns and the device mesh uploads; node_x / node_y
        # (contiguous, as the reference keeps them, ugrid2d.py:86-87) are cut out of it on first use
        self._node_xy = _node_table(node_x, node_y)
        self._node_x = self._node_y = None
        if not isinstance(face_node_connectivity, np.ndarray):
            raise TypeError("face_node_connectivity should be an array of integers")
        faces = _copy_connectivity(face_node_connectivity)
        if faces.ndim != 2:
            raise ValueError("face_node_connectivity must be 2-D (n_face, n_max_node_per_face)")
        # fill -> -1 and 0-based, as ugrid2d.py:105-110
        if fill_value != FILL_VALUE or start_index != 0:
            is_fill = faces == fill_value
            if start_index != 0:
                faces[~is_fill] -= start_index
            if fill_value != FILL_VALUE:
                faces[is_fill] = FILL_VALUE
        self.face_node_connectivity = faces
        self.fill_value = FILL_VALUE
        self.start_index = 0
        self.name = name
        self._celltree = None
        self._voronoi_device_cache = None  # (UnstructuredGrid2d._voronoi_device)
        self._area = None
        self._centroids = None
        self._edge_node_connectivity = None
        self._face_edge_connectivity = None
        self._edge_face_connectivity = None
        self._node_face_connectivity = None

    @property
    def node_x(self):
        if self._node_x is None:
            self._node_x = np.ascontiguousarray(self._node_xy[:, 0])
        return self._node_x

    @property
    def node_y(self):
        if self._node_y is None:
            self._node_y = np.ascontiguousarray(self._node_xy[:, 1])
        return self._node_y

    # plain attributes in the reference (ugrid2d.py:86-87): assigning new coordinates is legal there.  Here the interleaved
    # buffer is what the device sees, so an assignment rebuilds it and drops everything derived from the old coordinates
    # that THIS grid holds.  Regridders and UnstructuredGrid2d wrappers already built from the grid keep the weights they
    # computed from the old coordinates, as in the reference: build new ones.
    @node_x.setter
    def node_x(self, value):
        self._set_node_axis(0, value)

    @node_y.setter
    def node_y(self, value):
        self._set_node_axis(1, value)

    def _set_node_axis(self, axis, value):
        value = np.asarray(value, dtype=np.float64)
        if value.shape != (self.n_node,):
            raise ValueError(f"expected {self.n_node} node coordinates, got shape {value.shape}")
        xy = np.array(self._node_xy)  # (a fresh buffer: the old one may be in use by a device upload or a caller's view)
        xy[:, axis] = value
        self._node_xy = xy
        self._node_x = self._node_y = None
        self._area = self._centroids = None
        self.drop_device_caches()

    # ---- sizes / names
    @property
    def n_node(self):
        return self._node_xy.shape[0]

    @property
    def n_face(self):
        return self.face_node_connectivity.shape[0]

    @property
    def n_max_node_per_face(self):
        return self.face_node_connectivity.shape[1]

    @property
    def face_dimension(self):
        return f"{self.name}_nFaces"

    @property
    def core_dimension(self):
        return self.face_dimension

    @property
    def dims(self):
        return (self.face_dimension,)

    @property
    def node_coordinates(self):
        """(n_node, 2), a FRESH array per call as in the reference (``column_stack``, ugridbase.py:576-579): writing into it
        changes neither the grid nor its device copy -- assign ``node_x`` / ``node_y`` for that (the setters drop what was
        derived from the old coordinates).  Until round 5 this was a read-only view of the grid's own buffer; reference-style
        code that edits the returned array in place raised on it."""
        return self._node_xy.copy()

    @property
    def bounds(self):
        lo, hi = self._node_xy.min(axis=0), self._node_xy.max(axis=0)
        return (lo[0], lo[1], hi[0], hi[1])

    def node_coordinates_of(self, nodes):
        """(len(nodes), 2) coordinates of the given node ids."""
        return self._node_xy[nodes]

    # ---- device-backed geometry
    @property
    def celltree(self) -> CellTree2d:
        if self._celltree is None:
            self._celltree = CellTree2d(self._node_xy, self.face_node_connectivity, FILL_VALUE)
        return self._celltree

    def drop_device_caches(self):
        """Release what this grid keeps in HBM beyond its mesh: the celltree index and the cached centroidal Voronoi
        tessellation of the barycentric path (mesh, prepared arrays and index: several times ``n_face`` worth of memory that
        the engine's pool cannot reclaim while the grid is alive).  The next regridder on this grid rebuilds them."""
        self._celltree = None
        self._voronoi_device_cache = None

    @property
    def device_mesh(self):
        return self.celltree.device_mesh

    @property
    def area(self):
        if self._area is None:
            self._area = self.device_mesh.area()
        return self._area

    @property
    def centroids(self):
        if self._centroids is None:
            self._centroids = self.device_mesh.centroids()
        return self._centroids

    def locate_points(self, points, tolerance=None):
        return self.celltree.locate_points(points, tolerance)

    def compute_barycentric_weights(self, points, tolerance=None):
        return self.celltree.compute_barycentric_weights(points, tolerance)

    # ---- point sampling with the same spatial index (ugrid2d.py:1080-1140)
    def rasterize_like(self, x, y):
        """Face index at every (y, x) raster node: -> (x, y, index (nrow, ncol)), -1 outside the grid."""
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        # (the y.size * x.size sample points are generated on the device from the two 1-D arrays)
        return x, y, self.device_mesh.locate_raster(x, y)

    def rasterize(self, resolution, bounds=None):
        """Sample the grid on a raster of cell centres generated from ``bounds`` (default: the node bounds)
        and ``resolution``; y runs from top to bottom."""
        if bounds is None:
            bounds = self.bounds
        xmin, ymin, xmax, ymax = bounds
        d = abs(resolution)
        xmin = np.floor(xmin / d) * d
        xmax = np.ceil(xmax / d) * d
        ymin = np.floor(ymin / d) * d
        ymax = np.ceil(ymax / d) * d
        x = np.arange(xmin + 0.5 * d, xmax, d)
        y = np.arange(ymax - 0.5 * d, ymin, -d)
        return self.rasterize_like(x, y)

    # ---- host-side connectivities (feed the Voronoi pre-step of BarycentricInterpolator)
    @property
    def edge_node_connectivity(self):
        if self._edge_node_connectivity is None:
            self._edge_node_connectivity, self._face_edge_connectivity = connectivity.edge_connectivity(
                self.face_node_connectivity
            )
        return self._edge_node_connectivity

    @property
    def face_edge_connectivity(self):
        if self._face_edge_connectivity is None:
            self.edge_node_connectivity
        return self._face_edge_connectivity

    @property
    def edge_face_connectivity(self):
        if self._edge_face_connectivity is None:
            self._edge_face_connectivity = connectivity.invert_dense(self.face_edge_connectivity)
        return self._edge_face_connectivity

    @property
    def node_face_connectivity(self):
        if self._node_face_connectivity is None:
            self._node_face_connectivity = connectivity.invert_dense_to_sparse(
                self.face_node_connectivity, n_rows=self.n_node
            )
        return self._node_face_connectivity

    # ---- structured -> unstructured (raster cells become CCW quads)
    @staticmethod
    def _from_intervals_helper(node_x, node_y, nx, ny, name):
        # face id = row-major (y, x) in the bounds' own order; ugrid2d.py:1894-1912
        linear_index = np.arange(node_x.size, dtype=IntDType).reshape((ny + 1, nx + 1))
        face_nodes = np.empty((ny * nx, 4), dtype=IntDType)
        left, right = slice(None, -1), slice(1, None)
        lower, upper = slice(None, -1), slice(1, None)
        if node_x[1] < node_x[0]:
            left, right = right, left
        # NOTE: the reference tests `node_y[ny + 1] < node_y[0]` on the flattened vertex array
        # (ugrid2d.py:1906), which for nx > ny with descending y emits clockwise quads (SURVEY
        # appendix D).  Orientation does not matter downstream (the engine normalises every face
        # to CCW on the device), so the intended test -- first element of the second row -- is used.
        if node_y[nx + 1] < node_y[0]:
            lower, upper = upper, lower
        face_nodes[:, 0] = linear_index[lower, left].ravel()
        face_nodes[:, 1] = linear_index[lower, right].ravel()
        face_nodes[:, 2] = linear_index[upper, right].ravel()
        face_nodes[:, 3] = linear_index[upper, left].ravel()
        return Ugrid2d(node_x, node_y, FILL_VALUE, face_nodes, name=name)

    @staticmethod
    def from_structured_bounds(x_bounds, y_bounds, name="mesh2d"):
        """(nx, 2) and (ny, 2) cell bounds -> quad mesh; ugrid2d.py:1973-2034 (2-D bounds only)."""
        x_bounds = np.asarray(x_bounds, dtype=FloatDType)
        y_bounds = np.asarray(y_bounds, dtype=FloatDType)
        if x_bounds.ndim != 2 or y_bounds.ndim != 2:
            raise ValueError(f"Expected 2 dimensions on bounds, received: {x_bounds.ndim}")
        nx, ny = x_bounds.shape[0], y_bounds.shape[0]
        x = connectivity.bounds1d_to_vertices(x_bounds)
        y = connectivity.bounds1d_to_vertices(y_bounds)
        node_y, node_x = (a.ravel() for a in np.meshgrid(y, x, indexing="ij"))
        return Ugrid2d._from_intervals_helper(node_x, node_y, nx, ny, name)

    @staticmethod
    def from_structured_bounds_device(x_bounds, y_bounds, name="mesh2d"):
        """``from_structured_bounds`` with the mesh generated on the device: only the two 1-D vertex arrays are
        uploaded; the host copies of the node and face arrays are made on first access (persistence, host
        connectivities), never for regridding."""
        x_bounds = np.asarray(x_bounds, dtype=FloatDType)
        y_bounds = np.asarray(y_bounds, dtype=FloatDType)
        if x_bounds.ndim != 2 or y_bounds.ndim != 2:
            raise ValueError(f"Expected 2 dimensions on bounds, received: {x_bounds.ndim}")
        return RectilinearUgrid2d(
            connectivity.bounds1d_to_vertices(x_bounds), connectivity.bounds1d_to_vertices(y_bounds), name
        )

    # ---- persistence (plain dict of arrays; xarray is optional and absent here)
    def to_dataset(self, prefix=None):
        name = prefix if prefix is not None else self.name
        return {
            f"{name}_node_x": self.node_x,
            f"{name}_node_y": self.node_y,
            f"{name}_face_nodes": self.face_node_connectivity,
        }

    @staticmethod
    def from_device_arrays(node_coordinates, face_node_connectivity, fill_value=FILL_VALUE, name="mesh2d"):
        """A grid whose arrays ALREADY live in HBM: ``node_coordinates`` float64 ``(n_node, 2)`` and ``face_node_connectivity``
        int64 / int32 ``(n_face, n_max_node_per_face)`` as torch tensors on the GPU or anything with ``__cuda_array_interface__``
        (cupy, numba, ``engine.DeviceArray``).  Nothing crosses PCIe: the device mesh is made from the pointers
        (xr_mesh_create_dev validates and copies), the host arrays of the base class are downloaded only if somebody reads
        them.  (The reference has host grids only, ugrid2d.py:72-110; this is how its classes reach data a GPU pipeline
        already holds.)"""
        return DeviceUgrid2d(node_coordinates, face_node_connectivity, fill_value, name)

    @staticmethod
    def from_dataset(dataset, name):
        return Ugrid2d(
            np.asarray(dataset[f"{name}_node_x"]),
            np.asarray(dataset[f"{name}_node_y"]),
            FILL_VALUE,
            np.asarray(dataset[f"{name}_face_nodes"]),
            name=name,
        )


class RectilinearUgrid2d(Ugrid2d):
    """The quads of a rectilinear grid (``Ugrid2d.from_structured_bounds``, ugrid2d.py:1973-2034) whose node and
    face arrays exist on the DEVICE only: ``xr_mesh_create_rectilinear`` generates them from the two 1-D vertex
    arrays.  The host arrays of the base class are materialised lazily, on the first access of ``node_x`` /
    ``node_y`` / ``face_node_connectivity`` (persistence, host-side connectivities)."""

    def __init__(self, x_vertices, y_vertices, name="mesh2d"):
        self._xv = np.ascontiguousarray(x_vertices, dtype=FloatDType)
        self._yv = np.ascontiguousarray(y_vertices, dtype=FloatDType)
        if self._xv.ndim != 1 or self._yv.ndim != 1 or self._xv.size < 2 or self._yv.size < 2:
            raise ValueError("a rectilinear grid needs at least one cell per axis")
        self._host = None
        self.fill_value = FILL_VALUE
        self.start_index = 0
        self.name = name
        self._celltree = None
        self._area = None
        self._centroids = None
        self._edge_node_connectivity = None
        self._face_edge_connectivity = None
        self._edge_face_connectivity = None
        self._node_face_connectivity = None

    def _materialise(self):
        if self._host is None:
            node_y, node_x = (a.ravel() for a in np.meshgrid(self._yv, self._xv, indexing="ij"))
            self._host = Ugrid2d._from_intervals_helper(node_x, node_y, self._xv.size - 1, self._yv.size - 1, self.name)
        return self._host

    node_x = property(lambda self: self._materialise().node_x)
    node_y = property(lambda self: self._materialise().node_y)
    node_coordinates = property(lambda self: self._materialise().node_coordinates)
    _node_xy = property(lambda self: self._materialise()._node_xy)
    face_node_connectivity = property(lambda self: self._materialise().face_node_connectivity)

    @property
    def n_node(self):
        return self._xv.size * self._yv.size

    @property
    def n_face(self):
        return (self._xv.size - 1) * (self._yv.size - 1)

    @property
    def n_max_node_per_face(self):
        return 4

    @property
    def bounds(self):
        return (self._xv.min(), self._yv.min(), self._xv.max(), self._yv.max())

    def node_coordinates_of(self, nodes):
        nodes = np.asarray(nodes)
        j, i = np.divmod(nodes, self._xv.size)  # node id = j * (nx + 1) + i (meshgrid order)
        return np.column_stack([self._xv[i], self._yv[j]])

    @property
    def celltree(self) -> CellTree2d:
        if self._celltree is None:
            from .engine import DeviceMesh

            self._celltree = CellTree2d.from_device_mesh(DeviceMesh.from_rectilinear(self._xv, self._yv))
        return self._celltree


class DeviceUgrid2d(Ugrid2d):
    """``Ugrid2d.from_device_arrays``: the mesh exists on the device; host copies are made lazily (see RectilinearUgrid2d)."""

    def __init__(self, node_coordinates, face_node_connectivity, fill_value=FILL_VALUE, name="mesh2d"):
        from . import engine

        xy = engine.device_array_info(node_coordinates)
        faces = engine.device_array_info(face_node_connectivity)
        if xy is None or faces is None:
            raise TypeError("from_device_arrays expects device arrays (torch tensors on the GPU or __cuda_array_interface__)")
        (xy_ptr, xy_shape, xy_dtype), (f_ptr, f_shape, f_dtype) = xy, faces
        if len(xy_shape) != 2 or xy_shape[1] != 2 or xy_dtype != np.float64:
            raise ValueError("node_coordinates must be a float64 (n_node, 2) device array")
        if len(f_shape) != 2 or f_dtype not in (np.dtype(np.int64), np.dtype(np.int32)):
            raise ValueError("face_node_connectivity must be an int64 / int32 (n_face, n_max_node_per_face) device array")
        engine.sync_producer(node_coordinates)
        engine.sync_producer(face_node_connectivity)
        mesh = engine.DeviceMesh.from_device(xy_ptr, xy_shape[0], f_ptr, f_dtype.itemsize, f_shape[0], f_shape[1], fill_value)
        self._n_node, self._n_face, self._m = xy_shape[0], f_shape[0], f_shape[1]
        self._host = None
        self.fill_value = FILL_VALUE
        self.start_index = 0
        self.name = name
        self._celltree = CellTree2d.from_device_mesh(mesh)
        self._voronoi_device_cache = None
        self._area = None
        self._centroids = None
        self._edge_node_connectivity = None
        self._face_edge_connectivity = None
        self._edge_face_connectivity = None
        self._node_face_connectivity = None

    def _materialise(self):
        if self._host is None:
            xy, faces = self._celltree.device_mesh.download()
            self._host = Ugrid2d(xy[:, 0], xy[:, 1], FILL_VALUE, faces, name=self.name)
        return self._host

    node_x = property(lambda self: self._materialise().node_x)
    node_y = property(lambda self: self._materialise().node_y)
    node_coordinates = property(lambda self: self._materialise().node_coordinates)
    _node_xy = property(lambda self: self._materialise()._node_xy)
    face_node_connectivity = property(lambda self: self._materialise().face_node_connectivity)
    n_node = property(lambda self: self._n_node)
    n_face = property(lambda self: self._n_face)
    n_max_node_per_face = property(lambda self: self._m)
    bounds = property(lambda self: self._materialise().bounds)

    def node_coordinates_of(self, nodes):
        return self._materialise().node_coordinates_of(nodes)

    @property
    def celltree(self) -> CellTree2d:
        return self._celltree

    def drop_device_caches(self):
        # (the device mesh IS this grid: its derived arrays and index go, the raw arrays stay)
        self._voronoi_device_cache = None
        self._celltree.device_mesh.invalidate()
