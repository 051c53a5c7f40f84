"""
``CellTree2d``: the object ``Ugrid2d.celltree`` returns (xugrid/ugrid/ugrid2d.py:908-921), with the
method names, argument meaning and return conventions of ``numba_celltree.CellTree2d`` as xugrid
calls it -- so the same object can be handed to a real xugrid.  All geometry
runs in the HIP kernels behind ``xugrid_amd.engine.DeviceMesh``.
"""
import numpy as np

from . import engine
from .engine import DeviceMesh, IntDType


class CellTree2d:
    def __init__(self, vertices, faces, fill_value=-1, n_buckets=4, cells_per_leaf=2):
        # n_buckets / cells_per_leaf parameterise numba_celltree's bounding-box tree; the device
        # index is a hierarchical grid and has no such knobs.  Accepted for call compatibility.
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces)
        self.fill_value = fill_value
        self.n_buckets = n_buckets
        self.cells_per_leaf = cells_per_leaf
        self.device_mesh = DeviceMesh(self.vertices, self.faces, fill_value)

    @classmethod
    def from_device_mesh(cls, device_mesh: DeviceMesh, fill_value=-1):
        """Adapter around a mesh that already lives on the device (no host copy of its arrays is kept)."""
        self = cls.__new__(cls)
        self.vertices = self.faces = None
        self.fill_value = fill_value
        self.n_buckets, self.cells_per_leaf = 4, 2
        self.device_mesh = device_mesh
        return self

    def intersect_faces(self, vertices, faces, fill_value=-1):
        """
        Find all (query face, tree face) pairs with a positive intersection area
        (xugrid/regrid/unstructured.py:124-132).

        Returns ``(query_index, tree_index, area)`` ordered by query face (as MatrixCOO.to_csr
        requires, xugrid/core/sparse.py:65), tree faces ascending within a query face.
        """
        query = DeviceMesh(vertices, faces, fill_value)
        return self.intersect_mesh(query)

    def intersect_mesh(self, query: DeviceMesh, relative=False):
        csr = self.device_mesh.overlap(query, relative=relative)
        data, indices, indptr = csr.download()
        query_index = np.repeat(np.arange(csr.n, dtype=IntDType), np.diff(indptr))
        return query_index, indices, data

    def locate_points(self, points, tolerance=None):
        """Face index per point, -1 if none (unstructured.py:139,189; ugridbase.py:1323)."""
        return self.device_mesh.locate_points(points, tolerance)

    def compute_barycentric_weights(self, points, tolerance=None):
        """(face index [n], weights [n, n_max_node]) -- xugrid/ugrid/ugrid2d.py:1054-1078."""
        return self.device_mesh.compute_barycentric_weights(points, tolerance)

    def intersect_edges(self, edge_coords):
        """
        Intersections of line segments ``(n_edge, 2, 2)`` with the faces (xugrid/regrid/unstructured.py:203-215):
        ``(edge_index, face_index, intersections (n, 2, 2))`` ordered by edge, faces ascending within an edge.
        Only pieces of positive length are reported.
        """
        csr = engine.edge_length_csr(self.device_mesh, edge_coords)
        _, edge_index, indptr = csr.download()
        pieces = engine.edge_pieces(self.device_mesh, csr, edge_coords)
        face_index = np.repeat(np.arange(csr.n, dtype=IntDType), np.diff(indptr))
        order = np.lexsort((face_index, edge_index))
        return edge_index[order], face_index[order], pieces[order]
