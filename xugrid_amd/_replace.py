"""
``replace_interpolated_weights`` -- counterpart of xugrid/regrid/unstructured.py:17-57.

Moves the barycentric weight of a synthetic exterior Voronoi vertex (vertex id >=
``node_index_threshold``) onto its two projected neighbours by inverse distance.  Only points
inside exterior Voronoi cells are affected (a thin boundary layer), so the affected (point, slot)
entries are found vectorised and then processed in the reference's row-major order.
"""
import numpy as np


def replace_interpolated_weights(vertices, faces, face_index, weights, node_to_node_map, node_index_threshold):
    n, m = weights.shape
    if n == 0 or len(node_to_node_map) == 0:
        return
    valid = face_index >= 0
    rows = np.nonzero(valid)[0]
    if rows.size == 0:
        return
    face_rows = faces[face_index[rows]]  # (n_valid, m)
    substitute = face_rows >= node_index_threshold
    # candidate rows: a substitute vertex in the cell and some positive weight.  Inside a row the slots are walked in
    # the reference's order on the CURRENT weights (a slot may receive weight from an earlier one, or lose it)
    touched = np.nonzero(substitute.any(axis=1) & (weights[rows] > 0).any(axis=1))[0]
    ii, jj = np.nonzero(substitute[touched])
    for i_local, j in zip(touched[ii], jj):
        i = rows[i_local]
        face = face_rows[i_local]
        p = face[j]
        w = weights[i, j]
        if w <= 0:
            continue
        q, r = node_to_node_map[p - node_index_threshold]
        px, py = vertices[p]
        qx, qy = vertices[q]
        rx, ry = vertices[r]
        # explicit products: numba lowers ``x ** 2`` to a multiplication, libm's pow(x, 2.0) is not always the
        # correctly rounded square (found by the randomised soak test: 1 ulp in one weight of 749)
        p_q = np.sqrt((qx - px) * (qx - px) + (qy - py) * (qy - py))
        p_r = np.sqrt((rx - px) * (rx - px) + (ry - py) * (ry - py))
        total = p_q + p_r
        weight_q = (p_r / total) * w
        weight_r = (p_q / total) * w
        weights[i, j] = 0.0
        for k in range(m):
            node = face[k]
            if node == q:
                weights[i, k] += weight_q
            if node == r:
                weights[i, k] += weight_r
    return
