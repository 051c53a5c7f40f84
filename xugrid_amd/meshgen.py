"""
Synthetic meshes for the benchmark and the parity tests:
jittered-lattice points in the unit square, triangulated by scipy's Delaunay (qhull) when
available, else by splitting every jittered lattice quad along its shorter diagonal.  The
target meshes of the benchmark are the same generator with another seed, rotated 30 degrees
about (0.5, 0.5) and scaled by 0.7 so that they lie inside the source hull.
"""
import math

import numpy as np


def jittered_lattice_points(n_points, seed, jitter=0.35):
    """~n_points points: an m x m lattice on the unit square, jittered by U(-j, j) * h."""
    m = max(2, int(round(math.sqrt(n_points))))
    rng = np.random.default_rng(seed)
    h = 1.0 / (m - 1)
    gy, gx = np.meshgrid(np.arange(m) * h, np.arange(m) * h, indexing="ij")
    x = gx + rng.uniform(-jitter, jitter, gx.shape) * h
    y = gy + rng.uniform(-jitter, jitter, gy.shape) * h
    return np.column_stack([x.ravel(), y.ravel()]), m


def _split_lattice(points, m):
    """2 (m-1)^2 CCW triangles: each lattice quad split along its shorter diagonal (along the
    only valid diagonal where the jitter made the quad concave)."""
    idx = np.arange(m * m).reshape(m, m)
    a = idx[:-1, :-1].ravel()  # lower-left
    b = idx[:-1, 1:].ravel()  # lower-right
    c = idx[1:, 1:].ravel()  # upper-right
    d = idx[1:, :-1].ravel()  # upper-left
    p = points
    d_ac = ((p[a] - p[c]) ** 2).sum(axis=1)
    d_bd = ((p[b] - p[d]) ** 2).sum(axis=1)

    def ccw(i, j, k):
        u, v = p[j] - p[i], p[k] - p[i]
        return (u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]) > 0

    ok_ac = ccw(a, b, c) & ccw(a, c, d)
    ok_bd = ccw(a, b, d) & ccw(b, c, d)
    use_ac = ok_ac & (~ok_bd | (d_ac <= d_bd))
    t1 = np.where(use_ac[:, None], np.column_stack([a, b, c]), np.column_stack([a, b, d]))
    t2 = np.where(use_ac[:, None], np.column_stack([a, c, d]), np.column_stack([b, c, d]))
    faces = np.empty((2 * a.size, 3), dtype=np.int64)
    faces[0::2] = t1
    faces[1::2] = t2
    return faces


def triangle_mesh(n_points, seed, rotate_deg=0.0, scale=1.0, delaunay=True):
    """-> (node_xy float64[n,2], faces int64[F,3]); F ~ 2 n_points.  Faces are CCW."""
    points, m = jittered_lattice_points(n_points, seed)
    faces = None
    if delaunay:
        try:
            from scipy.spatial import Delaunay

            faces = Delaunay(points).simplices.astype(np.int64)
            # qhull orientation is not guaranteed: make every triangle CCW
            p = points
            u = p[faces[:, 1]] - p[faces[:, 0]]
            v = p[faces[:, 2]] - p[faces[:, 0]]
            cw = (u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]) < 0
            faces[cw] = faces[cw][:, ::-1]
        except ImportError:
            faces = None
    if faces is None:
        faces = _split_lattice(points, m)
    if rotate_deg != 0.0 or scale != 1.0:
        th = math.radians(rotate_deg)
        rot = np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])
        points = (points - 0.5) @ rot.T * scale + 0.5
    return np.ascontiguousarray(points), np.ascontiguousarray(faces)


def mixed_mesh(n_points, seed, rotate_deg=0.0, scale=1.0, quad_fraction=0.5, jitter=0.25):
    """A mixed triangle / quadrilateral mesh as flexible-mesh generators write them (dense (F, 4) connectivity, -1 in the
    fourth slot of a triangle): the quads of a jittered m x m lattice, a seeded share of them kept as (convex, CCW)
    quadrilaterals, the rest split into two CCW triangles along a valid diagonal.  -> (node_xy float64[n, 2], faces int64[F, 4]);
    F ~ (2 - quad_fraction) * n_points."""
    points, m = jittered_lattice_points(n_points, seed, jitter)
    idx = np.arange(m * m).reshape(m, m)
    a, b = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel()
    c, d = idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    p = points

    def ccw(i, j, k):
        u, v = p[j] - p[i], p[k] - p[i]
        return (u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]) > 0

    convex = ccw(a, b, c) & ccw(b, c, d) & ccw(c, d, a) & ccw(d, a, b)
    rng = np.random.default_rng(seed + 7919)
    keep = convex & (rng.random(a.size) < quad_fraction)
    tri = _split_lattice(points, m)  # two triangles per lattice quad, in quad order
    n_quad, n_split = int(keep.sum()), int((~keep).sum())
    faces = np.full((n_quad + 2 * n_split, 4), -1, dtype=np.int64)
    # faces in lattice order: a kept quad is one row, a split quad two
    rows = np.cumsum(np.where(keep, 1, 2)) - np.where(keep, 1, 2)
    faces[rows[keep]] = np.column_stack([a, b, c, d])[keep]
    split = np.nonzero(~keep)[0]
    faces[rows[split], :3] = tri[2 * split]
    faces[rows[split] + 1, :3] = tri[2 * split + 1]
    if rotate_deg != 0.0 or scale != 1.0:
        th = math.radians(rotate_deg)
        rot = np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])
        points = (points - 0.5) @ rot.T * scale + 0.5
    return np.ascontiguousarray(points), np.ascontiguousarray(faces)


def tiled_mesh(node_xy, faces, n_tiles):
    """n_tiles copies of a mesh side by side (tile t shifted by (t mod c, t div c) * 1.25 with c = ceil(sqrt(n_tiles))):
    the weak-scaling workload of the multi-GPU benchmark -- every tile is the single-GPU benchmark mesh, hull slivers
    included, and one qhull run serves any number of GPUs."""
    n_tiles = int(n_tiles)
    if n_tiles <= 1:
        return node_xy, faces
    c = int(math.ceil(math.sqrt(n_tiles)))
    xy = np.concatenate([node_xy + 1.25 * np.array([t % c, t // c], dtype=np.float64) for t in range(n_tiles)])
    f = np.concatenate([faces + t * node_xy.shape[0] for t in range(n_tiles)])
    return np.ascontiguousarray(xy), np.ascontiguousarray(f)


def quad_mesh(x_edges, y_edges):
    """Rectilinear quads: face id = row-major (y, x); CCW for ascending edges."""
    xe = np.asarray(x_edges, dtype=np.float64)
    ye = np.asarray(y_edges, dtype=np.float64)
    nx, ny = xe.size - 1, ye.size - 1
    yy, xx = np.meshgrid(ye, xe, indexing="ij")
    xy = np.column_stack([xx.ravel(), yy.ravel()])
    idx = np.arange((nx + 1) * (ny + 1)).reshape(ny + 1, nx + 1)
    faces = np.column_stack(
        [idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()]
    ).astype(np.int64)
    return xy, faces


def smooth_field(centroids, seed=0, nan_fraction=0.0):
    """v = sin(6 pi x) cos(4 pi y) + 0.1 N(0,1) at the given points (BASELINE.md C2 data)."""
    rng = np.random.default_rng(seed)
    x, y = centroids[:, 0], centroids[:, 1]
    v = np.sin(6 * np.pi * x) * np.cos(4 * np.pi * y) + 0.1 * rng.normal(size=x.size)
    if nan_fraction > 0:
        v[rng.random(x.size) < nan_fraction] = np.nan
    return v
