"""
Host-side (numpy) mesh algebra needed around the hot path -- the subset of
xugrid/ugrid/connectivity.py used by the regridders: polygon closing
(:372-382), dense connectivity inversion (:325-333), unique edges (:419-457) and the raster
bounds helper of xugrid/conversion.py:272-282.  Face areas and centroids are computed on the
device (xugrid_amd/csrc/xr_mesh.hip), not here.
"""
import numpy as np
from scipy import sparse

IntDType = np.intp
FILL_VALUE = -1


def bounds1d_to_vertices(bounds):
    """(n, 2) monotonic cell bounds -> n + 1 vertices in the bounds' own direction."""
    diff = np.diff(bounds, axis=0)
    if (diff >= 0.0).all():
        return np.concatenate((bounds[:, 0], bounds[-1:, 1]))
    if (diff <= 0.0).all():
        return np.concatenate((bounds[:, 1], bounds[-1:, 0]))
    raise ValueError("Bounds are not monotonic ascending or monotonic descending")


def close_polygons(face_node_connectivity):
    """Append node 0 and replace every fill slot by node 0; also return the fill mask."""
    n, m = face_node_connectivity.shape
    closed = np.full((n, m + 1), FILL_VALUE, dtype=IntDType)
    closed[:, :-1] = face_node_connectivity
    isfill = closed == FILL_VALUE
    first = np.broadcast_to(face_node_connectivity[:, :1], closed.shape)
    closed = np.where(isfill, first, closed)
    return closed, isfill


def _dense_to_coo(conn):
    n, m = conn.shape
    i = np.repeat(np.arange(n, dtype=IntDType), m)
    j = conn.ravel()
    valid = j != FILL_VALUE
    return i[valid], j[valid]


def invert_dense_to_sparse(conn, n_rows=None):
    """(a -> b) dense table with -1 fill  ->  scipy CSR of (b -> a), sorted column indices."""
    i, j = _dense_to_coo(conn)
    n_b = int(j.max()) + 1 if j.size else 0
    if n_rows is not None:
        n_b = max(n_b, n_rows)
    mat = sparse.coo_matrix((i, (j, i)), shape=(n_b, conn.shape[0])).tocsr()
    mat.sort_indices()
    return mat


def to_dense(mat):
    """scipy CSR -> dense table with -1 fill, entries in column order."""
    n = mat.shape[0]
    counts = np.diff(mat.indptr)
    m = int(counts.max()) if n else 0
    dense = np.full((n, m), FILL_VALUE, dtype=IntDType)
    rows = np.repeat(np.arange(n), counts)
    cols = np.arange(mat.indices.size) - np.repeat(mat.indptr[:-1], counts)
    dense[rows, cols] = mat.indices
    return dense


def invert_dense(conn):
    return to_dense(invert_dense_to_sparse(conn))


def edge_connectivity(face_node_connectivity):
    """
    Unique undirected edges of the faces.

    Returns ``edge_node_connectivity`` (n_edge, 2) with sorted node pairs in lexicographic
    order and ``face_edge_connectivity`` (n_face, n_max_node) with -1 fill.
    """
    n_face, n_max = face_node_connectivity.shape
    closed, isfill = close_polygons(face_node_connectivity)
    a = closed[:, :-1].ravel()
    b = closed[:, 1:].ravel()
    valid = a != b  # fill slots close onto node 0 -> self edges
    lo = np.minimum(a[valid], b[valid])
    hi = np.maximum(a[valid], b[valid])
    # unique undirected edges through ONE 1-D sort: key = lo * n_node + hi orders the edges
    # lexicographically, exactly like np.unique(pairs, axis=0) but ~10x faster
    n_node = int(max(lo.max(initial=-1), hi.max(initial=-1))) + 1
    key, inverse = np.unique(lo.astype(np.int64) * n_node + hi.astype(np.int64), return_inverse=True)
    edges = np.column_stack([key // max(n_node, 1), key % max(n_node, 1)])
    face_edges = np.full((n_face, n_max), FILL_VALUE, dtype=IntDType)
    face_edges.ravel()[np.nonzero(valid)[0]] = inverse.ravel()
    # compact each row to the left so that fill values trail
    order = np.argsort(face_edges == FILL_VALUE, axis=1, kind="stable")
    face_edges = np.take_along_axis(face_edges, order, axis=1)
    return edges.astype(IntDType), face_edges
