"""Shared inputs of the NetworkGridder tests (CPU oracle tests and GPU parity tests)."""
import numpy as np


def reference_case():
    """tests/test_regrid/test_network_gridder.py:12-72 of the reference: a 4 x 4 raster of unit cells (y descending),
    a network of four edges with data [1, 2, 4, -4], and the expected means at five sample cells."""
    raster = dict(y=np.arange(3.5, -0.5, -1.0), x=np.arange(0.5, 4.5, 1.0))
    node_xy = np.array([[0.0, 0.0], [1.5, 1.5], [2.5, 1.5], [4.0, 0.0], [4.0, 3.0]])
    edge_nodes = np.array([[0, 1], [1, 2], [2, 3], [2, 4]])
    data = np.array([1.0, 2.0, 4.0, -4.0])
    diag = 0.5 * np.sqrt(2)
    x_loc = np.array([0.5, 1.5, 2.5, 3.5, 3.5])
    y_loc = np.array([0.5, 1.5, 1.5, 2.5, 0.5])
    expected = np.array([
        1.0,
        (diag * 1 + 0.5 * 2) / (diag + 0.5),
        (0.5 * 2 + diag * -4 + diag * 4) / (2 * diag + 0.5),
        -4.0,
        4.0,
    ])
    return raster, node_xy, edge_nodes, data, (x_loc, y_loc, expected)


def raster_quads(x_edges, y_edges):
    """node_xy, faces (row-major over y then x) of the rectilinear grid with the given cell edges."""
    nx, ny = len(x_edges) - 1, len(y_edges) - 1
    X, Y = np.meshgrid(x_edges, y_edges)
    nodes = np.column_stack([X.ravel(), Y.ravel()])
    j, i = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    n0 = (j * (nx + 1) + i).ravel()
    faces = np.column_stack([n0, n0 + 1, n0 + nx + 2, n0 + nx + 1])
    return nodes, faces


def random_network(rng, n_edge, lo, hi, mean_len):
    """(n_edge, 2, 2) random segments whose first end point lies in [lo, hi]^2; lengths exponential."""
    a = rng.uniform(lo, hi, (n_edge, 2))
    ang = rng.uniform(0, 2 * np.pi, n_edge)
    length = rng.exponential(mean_len, n_edge)
    b = a + length[:, None] * np.column_stack([np.cos(ang), np.sin(ang)])
    return np.stack([a, b], axis=1)


def csr_from_pairs(edge_idx, face_idx, intersections, n_face):
    """The reference's post-processing (unstructured.py:211-215 + sparse.py:61-78) with edges ascending in a row."""
    d = np.diff(intersections, axis=1)[:, 0, :]
    length = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
    order = np.lexsort((edge_idx, face_idx))
    indptr = np.concatenate(([0], np.cumsum(np.bincount(face_idx, minlength=n_face)))).astype(np.int64)
    return length[order], edge_idx[order], indptr
