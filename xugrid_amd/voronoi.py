"""
Centroidal Voronoi topology of a mesh -- the host-side pre-step of ``BarycentricInterpolator``
(what xugrid/ugrid/voronoi.py:330-458 computes for the call made at
xugrid/regrid/unstructured.py:151-165, i.e. ``add_exterior=True, add_vertices=True,
skip_concave=True``; the other flag combinations are supported as well).

One Voronoi cell is built around every source NODE; its corners are the centroids of the faces
around that node, ordered counter-clockwise.  Cells of nodes on the mesh boundary are closed
with (a) the orthogonal projections of the adjacent face centroids onto the boundary edges and
(b) one extra corner per boundary node: the node itself when that keeps the cell convex,
otherwise the midpoint of its two projections.

The construction is expressed here as one flat table of (cell key, corner id) records that is
sorted once per group and packed into a dense, -1 padded connectivity.  Output conventions
(vertex numbering, cell order, record order) follow the reference so that cached weights and
goldens are interchangeable:

  vertices   = [face centroids ; edge projections ; per-boundary-node extra corners]
  cells      = interior nodes ascending, then boundary nodes ascending
  face_index = source face of each vertex, -1 for the extra corners
  interpolation_map[k] = the two projection vertex ids the k-th extra corner sits between
"""
import numpy as np
import scipy.sparse

from .connectivity import FILL_VALUE, IntDType, close_polygons

_MERGE_TOL = 1.0e-8 * 1.0e-8  # projections closer than this to their centroid are dropped


def _pack_rows(keys, values):
    """Records already grouped by ascending key -> dense (n_groups, max_len) table, -1 padded."""
    counts = np.bincount(keys)
    counts = counts[counts > 0]
    n, m = counts.size, int(counts.max())
    table = np.full((n, m), FILL_VALUE, dtype=IntDType)
    starts = np.zeros(n, dtype=IntDType)
    np.cumsum(counts[:-1], out=starts[1:])
    col = np.arange(values.size) - np.repeat(starts, counts)
    table[np.repeat(np.arange(n), counts), col] = values
    return table


def _fan_area(xy):
    """0.5 |sum cross(p_k - p_0, p_{k+1} - p_0)| for closed polygons xy[(n, m + 1, 2)]."""
    rel = xy - xy[:, :1]
    a, b = rel[:, :-1], rel[:, 1:]
    return 0.5 * np.abs((a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]).sum(axis=1))


def _group_mean(keys, x, y):
    """Mean of (x, y) per distinct key, returned aligned with the records."""
    uniq, inv = np.unique(keys, return_inverse=True)
    cnt = np.bincount(inv).astype(np.float64)
    mx = np.bincount(inv, weights=x) / cnt
    my = np.bincount(inv, weights=y) / cnt
    return mx[inv], my[inv]


def _ccw_sort(keys, corner_ids, corner_xy, pivot_x, pivot_y):
    """Order the records by (key, angle of corner about its pivot)."""
    angle = np.arctan2(corner_xy[:, 1] - pivot_y, corner_xy[:, 0] - pivot_x)
    order = np.lexsort((angle, keys))
    return keys[order], corner_ids[order]


def _boundary_records(nfc, node_xy, centroids, edge_nodes, edge_face, add_vertices, skip_concave):
    """Cells of the boundary nodes from the exterior edges (edge_nodes (n_be, 2), edge_face (n_be,)).
    -> (vertex table, keys, corner ids, face_index, interp map)"""
    n_face = nfc.shape[1]
    per_node = np.diff(nfc.indptr)

    # -- corners that are face centroids
    bnodes = np.unique(edge_nodes.ravel())
    multi = bnodes[per_node[bnodes] > 1]            # boundary nodes shared by several faces
    sel = nfc[multi]
    keys_a = np.repeat(multi, np.diff(sel.indptr))
    ids_a = sel.indices
    single = np.nonzero(per_node == 1)[0]           # corner nodes owned by exactly one face
    keys_b = single
    ids_b = nfc[single].indices

    # -- corners that are projections of a centroid on a boundary edge
    a = node_xy[edge_nodes[:, 0]]
    b = node_xy[edge_nodes[:, 1]]
    c = centroids[edge_face]
    v, u = b - a, c - a
    s = (u[:, 0] * v[:, 0] + u[:, 1] * v[:, 1]) / (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1])
    proj_all = a + s[:, None] * v
    keep = np.linalg.norm(proj_all - c, axis=1) > _MERGE_TOL
    proj = proj_all[keep]
    n_proj = proj.shape[0]
    first_new = n_face + n_proj
    keys_c = edge_nodes[keep].ravel()               # both end nodes use the projection
    ids_c = np.repeat(np.arange(n_face, first_new), 2)
    vertex_face = edge_face[keep]
    extra_xy = np.zeros((0, 2))
    interp_map = None
    n_extra = 0
    keys_d = ids_d = np.zeros(0, dtype=IntDType)
    if add_vertices:
        # one extra corner per boundary node, between the node's two projections.  The pairing
        # walks the (edge, end) records sorted by node id; ids refer to the unfiltered projection
        # numbering exactly as the reference does.
        flat_nodes = edge_nodes.ravel()
        by_node = np.argsort(flat_nodes, kind="stable")
        pair_ids = np.repeat(np.arange(proj_all.shape[0]), 2)[by_node]
        pair_xy = proj_all[pair_ids]
        extra_xy = 0.5 * (pair_xy[0::2] + pair_xy[1::2])
        n_extra = extra_xy.shape[0]
        keys_d = flat_nodes[by_node][0::2]
        ids_d = np.arange(first_new, first_new + n_extra)
        interp_map = pair_ids.reshape(-1, 2) + n_face
        vertex_face = np.concatenate([vertex_face, np.full(n_extra, -1, dtype=vertex_face.dtype)])

    keys = np.concatenate([keys_a, keys_b, keys_c, keys_d]).astype(IntDType)
    ids = np.concatenate([ids_a, ids_b, ids_c, ids_d]).astype(IntDType)
    table = np.concatenate([centroids, proj, extra_xy])
    face_index = np.concatenate([np.arange(n_face), vertex_face])
    true_corner = node_xy[keys_d] if n_extra else np.zeros((0, 2))

    # -- counter-clockwise order about the mean of each cell's corners
    xy = table[ids]
    px, py = _group_mean(keys, xy[:, 0], xy[:, 1])
    keys, ids = _ccw_sort(keys, ids, xy, px, py)

    if add_vertices and n_extra:
        if skip_concave:
            # keep the true boundary node where it does not make the cell concave: compare the
            # cell area with the midpoint substitute against the area with the true node
            cells = _pack_rows(keys, ids)
            closed, _ = close_polygons(cells)
            swapped = table.copy()
            swapped[-n_extra:] = true_corner
            use_true = _fan_area(swapped[closed]) >= _fan_area(table[closed])
            is_extra = cells >= table.shape[0] - n_extra
            chosen = cells[use_true[:, None] & is_extra]
            table[chosen] = swapped[chosen]
        else:
            table[-n_extra:] = true_corner
    return table, keys, ids, face_index, interp_map


def voronoi_topology(
    node_face_connectivity,
    vertices,
    centroids,
    edge_face_connectivity=None,
    edge_node_connectivity=None,
    add_exterior=False,
    add_vertices=False,
    skip_concave=False,
):
    """
    Returns
    -------
    nodes: (n_vertex, 2) floats
    face_node_connectivity: (n_cell, n_max) ints, -1 padded, corners counter-clockwise
    face_index: (n_vertex,) source face of every Voronoi vertex (-1: extra boundary corner)
    interpolation_map: (n_extra, 2) ints or None
    """
    nfc = node_face_connectivity.tocsr()
    vertices = np.asarray(vertices, dtype=np.float64)
    centroids = np.asarray(centroids, dtype=np.float64)
    if add_exterior and (edge_face_connectivity is None or edge_node_connectivity is None):
        raise ValueError(
            "edge_face_connectivity, edge_node_connectivity must be provided if add_exterior is True."
        )
    per_node = np.diff(nfc.indptr)
    n_node = nfc.shape[0]
    if add_exterior:
        touches_boundary = np.zeros(max(n_node, vertices.shape[0]), dtype=bool)
        ext_edge = edge_face_connectivity[:, 1] == FILL_VALUE
        touches_boundary[edge_node_connectivity[ext_edge].ravel()] = True
        node_ok = ~touches_boundary[:n_node]
    else:
        node_ok = per_node >= 3
    rec_ok = np.repeat(node_ok, per_node)
    keys = np.repeat(np.arange(n_node, dtype=IntDType), per_node)[rec_ok]
    ids = nfc.indices[rec_ok].astype(IntDType)
    pivots = vertices[keys]
    keys, ids = _ccw_sort(keys, ids, centroids[ids], pivots[:, 0], pivots[:, 1])

    if add_exterior:
        table, bkeys, bids, face_index, interp_map = _boundary_records(
            nfc, vertices, centroids, edge_node_connectivity[ext_edge], edge_face_connectivity[ext_edge, 0],
            add_vertices, skip_concave,
        )
        shift = int(keys.max()) + 1 if keys.size else 0
        keys = np.concatenate([keys, bkeys + shift])
        ids = np.concatenate([ids, bids])
    else:
        interp_map = None
        used = np.unique(ids)
        table = centroids[used]
        face_index = np.arange(int(ids.max()) + 1) if ids.size else np.zeros(0, dtype=IntDType)
        ids = np.searchsorted(used, ids)
    cells = _pack_rows(keys, ids)
    return table, cells, face_index, interp_map


def voronoi_topology_device(grid, compact=False, host_boundary=False):
    """
    ``voronoi_topology(..., add_exterior=True, add_vertices=True, skip_concave=True)`` of a Ugrid2d with the
    O(n) part on the device (node -> face inversion, exterior edges, counter-clockwise interior cells, assembly;
    xugrid_amd/csrc/xr_voronoi.hip) and only the cells of the boundary nodes on the host, computed on a LOCAL
    problem (boundary nodes, the faces around them) from a few KB the device gathers -- nothing of size O(n)
    crosses PCIe or is touched by numpy.

    Returns (DeviceMesh of the tessellation, face_index, interpolation_map): the mesh has the same vertices and
    cells, in the same order, as the host function returns as arrays.  ``compact=True``: ``face_index`` holds only
    the entries of the vertices beyond the ``n_face`` face centroids (the others are the identity).

    The cells of the boundary nodes are computed by the library itself (native code, ``xr_voronoi_mesh_auto``);
    ``host_boundary=True`` uses the numpy restatement below instead (kept as the readable cross-check: both give the
    same vertices, cells, face index and interpolation map bit for bit).
    """
    from . import engine

    builder = engine.DeviceVoronoi(grid.device_mesh)
    n_face = grid.n_face
    if not host_boundary:
        mesh, tail, interp_map = builder.assemble_auto()
        tail = tail.astype(IntDType, copy=False)
        interp_map = interp_map.astype(IntDType, copy=False)
        if compact:
            return mesh, tail, interp_map
        return mesh, np.concatenate([np.arange(n_face, dtype=IntDType), tail]), interp_map
    nodes, row_ptr, faces, face_xy, edge_nodes, edge_face, edge_face_xy = builder.download_boundary()
    if edge_face.size:
        # local numbering: boundary nodes 0..nb-1 (ascending = same order as their global ids), needed faces
        # 0..nl-1 (ascending), so every sort and grouping below sees the order it would see globally
        needed, inverse = np.unique(np.concatenate([faces, edge_face]), return_inverse=True)
        nl = needed.size
        cen = np.empty((nl, 2))
        cen[inverse[: faces.size]] = face_xy
        cen[inverse[faces.size:]] = edge_face_xy
        nfc = scipy.sparse.csr_matrix(
            (np.ones(faces.size, dtype=np.int8), inverse[: faces.size], row_ptr), shape=(nodes.size, nl)
        )
        node_xy = grid.node_coordinates_of(nodes)
        table, bkeys, bids, findex, interp_map = _boundary_records(
            nfc, node_xy, cen, np.searchsorted(nodes, edge_nodes), inverse[faces.size:], True, True
        )
        shift = n_face - nl  # local id of an added vertex -> its global id
        bids = np.where(bids < nl, needed[np.minimum(bids, nl - 1)], bids + shift)
        cells = _pack_rows(bkeys, bids)  # (local keys: same grouping and order as the global node ids)
        extra = table[nl:]
        tail = findex[nl:]
        tail = np.where(tail >= 0, needed[np.maximum(tail, 0)], -1).astype(IntDType)
        interp_map = interp_map + shift
    else:  # closed surface: nothing to add
        cells = np.zeros((0, 3), dtype=IntDType)
        extra = np.zeros((0, 2))
        tail, interp_map = np.zeros(0, dtype=IntDType), np.zeros((0, 2), dtype=IntDType)
    mesh = builder.assemble(extra, cells)
    if compact:
        return mesh, tail, interp_map
    return mesh, np.concatenate([np.arange(n_face, dtype=IntDType), tail]), interp_map
