"""
Test infrastructure: the reference's barycentric / centroid-locator weight construction STEP BY STEP
(xugrid/regrid/unstructured.py:137-201), once with the CPU oracle behind every call and once with the product's
single-purpose entry points (compute_barycentric_weights, replace_interpolated_weights, locate_points) chained on the
host.  The product itself builds these weights in one device pass (UnstructuredGrid2d.barycentric_device /
locate_centroids_device); the tests compare that pass with both restatements.  Until round 5 the second one lived in
the product package (xugrid_amd/regrid/unstructured.py) although only tests called it.
"""
import numpy as np


def oracle_barycentric_triplets(oracle, grid, points, tolerance=None, tree_order=False):
    """unstructured.py:146-201 with the CPU oracle (Voronoi pre-step: xugrid_amd.voronoi, pinned by golden G6).
    The weight slots are paired with the CALLER's vertex order of every Voronoi cell, as :175,193 do (default);
    ``tree_order``: with the tree's counter-clockwise-normalised copy instead (the product's opt-in)."""
    from xugrid_amd import voronoi

    xy = grid.node_coordinates
    faces = grid.face_node_connectivity
    vertices, vfaces, node_to_face_index, n2n = voronoi.voronoi_topology(
        grid.node_face_connectivity, xy, oracle.centroids(xy, faces),
        edge_face_connectivity=grid.edge_face_connectivity, edge_node_connectivity=grid.edge_node_connectivity,
        add_exterior=True, add_vertices=True, skip_concave=True,
    )
    vtree = oracle.CellTree2d(vertices, vfaces, -1)
    face_index, weights = vtree.compute_barycentric_weights(points, tolerance)
    pair_faces = vtree.faces if tree_order else np.asarray(vfaces)
    assert pair_faces.shape == vtree.faces.shape
    oracle.replace_interpolated_weights(vertices, pair_faces, face_index, weights, n2n, len(vertices) - len(n2n))
    outside = oracle.CellTree2d(xy, faces, -1).locate_points(points) == -1
    weights[outside] = 0
    keep = weights.ravel() > 0
    source_index = node_to_face_index[pair_faces[face_index]].ravel()[keep]
    n, m = weights.shape
    target_index = np.repeat(np.arange(n), m)[keep]
    return source_index, target_index, weights.ravel()[keep]


def host_barycentric_stepwise(us, ut, tolerance=None, tree_order=False):
    """The same steps through the product's single-purpose device calls, glued on the host: -> (source_index,
    target_index, weights).  ``us`` / ``ut``: xugrid_amd.regrid.UnstructuredGrid2d."""
    from xugrid_amd._replace import replace_interpolated_weights

    points = ut.ugrid_topology.centroids
    grid = us.ugrid_topology
    voronoi_grid, vertices, faces, node_to_face_index, node_to_node_map = us._voronoi()
    face_index, weights = voronoi_grid.compute_barycentric_weights(points, tolerance)
    if tree_order:
        faces = voronoi_grid.device_mesh.faces_ccw()
    replace_interpolated_weights(
        vertices=vertices, faces=faces, face_index=face_index, weights=weights,
        node_to_node_map=node_to_node_map, node_index_threshold=len(vertices) - len(node_to_node_map),
    )
    outside = grid.locate_points(points) == -1
    weights[outside] = 0
    keep = weights.ravel() > 0
    source_index = node_to_face_index[faces[face_index]].ravel()[keep]
    n_points, n_max_node = weights.shape
    target_index = np.repeat(np.arange(n_points, dtype=np.int64), n_max_node)[keep]
    return source_index, target_index, weights.ravel()[keep]


def host_locate_centroids_stepwise(us, ut, tolerance=None):
    """unstructured.py:137-144 through ``locate_points``."""
    source_index = us.ugrid_topology.celltree.locate_points(ut.ugrid_topology.centroids, tolerance)
    inside = source_index != -1
    source_index = source_index[inside]
    target_index = np.arange(ut.size, dtype=np.int64)[inside]
    return source_index, target_index, np.ones(source_index.size)


# the reference's test_barycentric_concave (tests/test_regrid/test_regridder.py:334-369), numbers transcribed:
# three triangles around a reflex corner, interpolated onto a 30 x 20 raster of 0.1-wide cells
CONCAVE_VERTICES = np.array([[0.0, 0.0], [3.0, 0.0], [1.0, 1.0], [0.0, 2.0], [3.0, 2.0]])
CONCAVE_FACES = np.array([[0, 1, 2], [0, 2, 3], [2, 4, 3]])
CONCAVE_VALUES = np.array([2.0, 0.5, 2.0])
CONCAVE_DX = 0.1


def concave_raster_axes():
    x = np.arange(0.0, 3.0, CONCAVE_DX) + 0.5 * CONCAVE_DX
    y = np.arange(0.0, 2.0, CONCAVE_DX) + 0.5 * CONCAVE_DX
    return x, y
