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


def line_selection_cases():
    """Known answers of the reference's line selections, which are numba_celltree.intersect_edges results
    post-processed by selection_utils.py:27-32 / ugridbase.py:1371-1450 (piece midpoints, distance along the line):
    tests/test_ugrid2d.py:1153-1190 and tests/test_ugrid_dataset.py:255-281 on the seven-node mesh of two unit
    quads below two triangles.  Returns (nodes, faces, cases); a case is (segments (n, 2, 2), face ids, mid x,
    mid y, s) in the order of increasing s."""
    nodes = np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0], [0.0, 1.0], [1.0, 1.0], [2.0, 1.0], [1.0, 2.0]])
    faces = np.array([[0, 1, 4, 3], [1, 2, 5, 4], [3, 4, 6, -1], [4, 5, 6, -1]])
    r2 = np.sqrt(2.0)
    diagonal = (np.array([[[0.0, 0.0], [2.0, 2.0]]]), [0, 3], [0.5, 1.25], [0.5, 1.25], [0.5 * r2, 1.25 * r2])
    # faces 1 and 2 are touched at the corner (1, 1) only: no piece.  The bend (1.5, 1.5) lies ON the hypotenuse of face 3.
    bend = (np.array([[[0.5, 0.5], [1.5, 0.5]], [[1.5, 0.5], [1.5, 1.5]]]), [0, 1, 1, 3], [0.75, 1.25, 1.5, 1.5],
            [0.5, 0.5, 0.75, 1.25], [0.25, 0.75, 1.25, 1.75])
    # ugrid2d.sel with a full slice in x and y = 0.5 (and x = 0.5, all y): tests/test_ugrid2d.py:1120-1145; the line
    # spans the bounding box of the mesh
    along_x = (np.array([[[0.0, 0.5], [2.0, 0.5]]]), [0, 1], [0.5, 1.5], [0.5, 0.5], [0.5, 1.5])
    along_y = (np.array([[[0.5, 0.0], [0.5, 2.0]]]), [0, 2], [0.5, 0.5], [0.5, 1.25], [0.5, 1.25])
    return nodes, faces, (diagonal, bend, along_x, along_y)


def burn_lines_case():
    """tests/test_burn.py:20-27,44-62,135-141 of the reference: _burn_lines (burn.py:153-181) writes ``values[line]`` into every
    face ``intersect_edges`` reports for a segment of the line, on a 3 x 3 grid of unit quads (face id = 3 row + column).
    Returns (nodes, faces, segments (5, 2, 2), value per segment, expected output with -1 for untouched faces)."""
    gy, gx = np.meshgrid(np.arange(4.0), np.arange(4.0), indexing="ij")
    nodes = np.column_stack([gx.ravel(), gy.ravel()])
    v = (4 * np.arange(3)[:, None] + np.arange(3)[None, :]).ravel()
    faces = np.column_stack([v, v + 1, v + 5, v + 4])
    xy = np.array([[0.5, 0.5], [2.5, 0.5], [1.2, 1.5], [1.8, 1.5], [0.2, 2.2], [0.8, 2.8], [1.2, 2.2], [1.8, 2.8]])
    segments = np.array([[xy[0], xy[1]], [xy[2], xy[3]], [xy[4], xy[5]], [xy[5], xy[6]], [xy[6], xy[7]]])
    values = np.array([0.0, 1.0, 2.0, 2.0, 2.0])
    expected = np.array([0.0, 0.0, 0.0, -1.0, 1.0, -1.0, 2.0, 2.0, -1.0])
    return nodes, faces, segments, values, expected


def line_selection_of_pairs(segments, edge_idx, face_idx, intersections):
    """selection_utils.py:27-32 + ugridbase.py:1412-1450 on intersect_edges output: (faces, mid x, mid y, s) by s."""
    mid = 0.5 * (intersections[:, 0, :] + intersections[:, 1, :])
    seg_len = np.hypot(*(segments[:, 1] - segments[:, 0]).T)
    before = np.concatenate(([0.0], np.cumsum(seg_len)[:-1]))
    s = np.hypot(*(mid - segments[edge_idx, 0]).T) + before[edge_idx]
    order = np.argsort(s, kind="stable")
    return face_idx[order], mid[order, 0], mid[order, 1], s[order]


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
