#!/usr/bin/env python
"""
numba_celltree CONFORMANCE KIT -- turns the "assumed numba_celltree behaviour" table of DESIGN.md section 7 into one command.

    python tests/golden/make_g11_celltree.py [output.npz]        (default: g11_celltree.npz next to this script)

Needs ONLY numpy and numba_celltree (the version xugrid pins: 0.4.x, pixi.lock:298); it imports nothing from this repository
and nothing from xugrid.  IT CANNOT BE RUN IN THE BUILD IMAGE of this repository: numba_celltree (and numba) are not installed
there and there is no network -- which is exactly why the clip / locate / barycentric / edge arithmetic of the package is
"parity unpinned" (DESIGN.md section 4).  Run it on any machine that has the package, drop the file it writes into
tests/golden/, and `pytest tests/test_celltree_conformance.py` compares the CPU oracle (always) and the HIP engine (-m gpu) with
what the REAL package returned, case by case; without the file those tests are skipped with a loud reason.

Every case stores its INPUTS next to the package's OUTPUTS, so the test side regenerates nothing.  The calls are the reference's
own call sites:
    CellTree2d(vertices, faces, fill_value)                       xugrid/ugrid/ugrid2d.py:915-921
    tree.intersect_faces(vertices, faces, fill_value)             xugrid/regrid/unstructured.py:124-132
    tree.locate_points(points, tolerance)                         xugrid/regrid/unstructured.py:139,189; ugridbase.py:1323
    tree.compute_barycentric_weights(points, tolerance)           xugrid/ugrid/ugrid2d.py:1078
    tree.intersect_edges(edge_coords)                             xugrid/regrid/unstructured.py:203-215

Cases (names are the npz key prefixes):
  faces_self, faces_self_utm      a triangle mesh against ITSELF (every neighbour shares nodes; corner neighbours only touch), near the
                                  origin and at UTM-like coordinates: strict box test + SAT before the clip, `area > 0`
  faces_subset                    a coarse triangulation of a SUBSET of the fine mesh's nodes against the fine mesh, both ways
  faces_needles                   sources with needle-thin (height ~1e-13) and zero-area (repeated node) triangles
  faces_quads_self, faces_tri_quad  quadrilaterals against themselves; triangles against quadrilaterals with fill values
  faces_general                   a rotated, scaled, differently seeded target: GENERAL triangle x triangle areas (no reference test
                                  pins one)
  locate_ties                     points ON shared sides, ON nodes, ON the hull, just outside: which face wins a tie, the default
                                  tolerance and an explicit one
  locate_tolform                  a long and a short hull side, probes at 0.9 / 1.1 x the tolerance outside them: tells
                                  `cross < tol * length` (an absolute distance) from `cross < tol`
  bary_concave                    polygons incl. a concave one that STARTS at its reflex corner (the counter-clockwise
                                  normalisation by the first vertex triple) -- barycentric weights slot by slot
  edges_touch                     segments through a cell corner, along a shared side, ending on a side, outside along the hull
"""
import os
import sys

import numpy as np


def jittered_lattice(m, seed, jitter=0.35):
    rng = np.random.default_rng(seed)
    h = 1.0 / (m - 1)
    gy, gx = np.meshgrid(np.arange(m) * h, np.arange(m) * h, indexing="ij")
    x = gx + rng.uniform(-jitter, jitter, gx.shape) * h
    y = gy + rng.uniform(-jitter, jitter, gy.shape) * h
    return np.column_stack([x.ravel(), y.ravel()])


def split_lattice(p, m, step=1):
    """CCW triangles of the m x m lattice nodes taken every `step`: each quad split along a valid diagonal."""
    idx = np.arange(m * m).reshape(m, m)[::step, ::step]
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()

    def ccw(i, j, k):
        u, v = p[j] - p[i], p[k] - p[i]
        return (u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]) > 0

    ok_ac = ccw(a, b, c) & ccw(a, c, d)
    ok_bd = ccw(a, b, d) & ccw(b, c, d)
    d_ac = ((p[a] - p[c]) ** 2).sum(axis=1)
    d_bd = ((p[b] - p[d]) ** 2).sum(axis=1)
    use_ac = ok_ac & (~ok_bd | (d_ac <= d_bd))
    t1 = np.where(use_ac[:, None], np.column_stack([a, b, c]), np.column_stack([a, b, d]))
    t2 = np.where(use_ac[:, None], np.column_stack([a, c, d]), np.column_stack([b, c, d]))
    faces = np.empty((2 * a.size, 3), dtype=np.int64)
    faces[0::2], faces[1::2] = t1, t2
    return faces


def quads(nx, ny):
    xe, ye = np.linspace(0.0, 1.0, nx + 1), np.linspace(0.0, 1.0, ny + 1)
    yy, xx = np.meshgrid(ye, xe, indexing="ij")
    xy = np.column_stack([xx.ravel(), yy.ravel()])
    idx = np.arange((nx + 1) * (ny + 1)).reshape(ny + 1, nx + 1)
    f = np.column_stack([idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()]).astype(np.int64)
    return xy, f


def rotated(p, deg, scale):
    th = np.radians(deg)
    rot = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    return (p - 0.5) @ rot.T * scale + 0.5


def main(path):
    try:
        import numba_celltree
        from numba_celltree import CellTree2d
    except ImportError as e:  # the expected outcome inside this repository's build image
        sys.exit(f"numba_celltree is not importable here ({e}).  This kit has to run on a machine that has the package; "
                 "see the header of this script.")

    out = {"_numba_celltree_version": np.array(getattr(numba_celltree, "__version__", "unknown"))}

    def faces_case(name, sxy, sf, txy, tf, fill=-1):
        tree = CellTree2d(np.ascontiguousarray(sxy), np.ascontiguousarray(sf), fill)
        q, s, area = tree.intersect_faces(np.ascontiguousarray(txy), np.ascontiguousarray(tf), fill)
        order = np.lexsort((s, q))
        out.update({f"{name}__sxy": sxy, f"{name}__sf": sf, f"{name}__txy": txy, f"{name}__tf": tf, f"{name}__fill": np.array(fill),
                    f"{name}__q": np.asarray(q)[order], f"{name}__s": np.asarray(s)[order], f"{name}__area": np.asarray(area)[order]})
        print(f"{name}: {len(q)} pairs")

    m = 61
    p = jittered_lattice(m, 21)
    f = split_lattice(p, m)
    faces_case("faces_self", p, f, p, f)
    utm = p * 3000.0 + np.array([6.5e5, 5.9e6])
    faces_case("faces_self_utm", utm, f, utm, f)
    # coarse faces on every third lattice node: an exact subset of the fine nodes (renumbered)
    coarse_ids = np.arange(m * m).reshape(m, m)[::3, ::3]
    cf_global = split_lattice(p, m, 3)
    remap = -np.ones(m * m, dtype=np.int64)
    remap[coarse_ids.ravel()] = np.arange(coarse_ids.size)
    cp, cf = p[coarse_ids.ravel()], remap[cf_global]
    faces_case("faces_subset_fine_tree", p, f, cp, cf)
    faces_case("faces_subset_coarse_tree", cp, cf, p, f)
    # needle / zero-area sources
    rng = np.random.default_rng(5)
    nxy, f2 = p.copy(), f.copy()
    for k, face in enumerate(rng.choice(f.shape[0], 200, replace=False)):
        a, b, c = f2[face]
        if k % 2 == 0:
            nxy = np.vstack([nxy, 0.5 * (nxy[a] + nxy[b]) + 1e-13 * (nxy[c] - nxy[a])])
            f2[face] = [a, b, nxy.shape[0] - 1]
        else:
            f2[face] = [a, b, b]
    tp = rotated(jittered_lattice(51, 22), 25.0, 0.8)
    tf = split_lattice(tp, 51)
    faces_case("faces_needles", nxy, f2, tp, tf)
    faces_case("faces_needles_self", nxy, f2, p, f)
    qxy, qf = quads(37, 29)
    faces_case("faces_quads_self", qxy, qf, qxy, qf)
    tri_fill = np.column_stack([f, np.full(f.shape[0], -1, dtype=np.int64)])
    faces_case("faces_tri_quad", qxy, qf, p, tri_fill)       # triangles stored with a fill column against a quad tree
    faces_case("faces_quad_tri", p, f, qxy, qf)
    faces_case("faces_general", p, f, tp, tf)                # general position: rotated 25 degrees, scaled 0.8, other seed
    g2 = rotated(jittered_lattice(90, 31), 30.0, 0.7)
    faces_case("faces_general_fine", p, f, g2, split_lattice(g2, 90))

    # ---- locate_points: ties
    lm = 12
    lp = jittered_lattice(lm, 3, jitter=0.2)
    lf = split_lattice(lp, lm)
    tree = CellTree2d(lp, lf, -1)
    e = np.vstack([lf[:, [0, 1]], lf[:, [1, 2]], lf[:, [2, 0]]])
    mid = 0.5 * (lp[e[:, 0]] + lp[e[:, 1]])                      # side midpoints: interior sides appear twice (both faces)
    third = lp[e[:, 0]] + (lp[e[:, 1]] - lp[e[:, 0]]) / 3.0
    hull_out = np.array([[-1e-12, 0.5], [0.5, -1e-12], [1.0 + 1e-9, 0.3], [0.3, 1.0 + 1e-3], [-0.5, -0.5]])
    pts = np.vstack([lp, mid, third, lp[lf].mean(axis=1), hull_out])
    out.update({"locate_ties__xy": lp, "locate_ties__faces": lf, "locate_ties__points": pts})
    try:
        out["locate_ties__default"] = np.asarray(tree.locate_points(pts, None))
        for tol in (1e-12, 1e-6, 1e-2):
            out[f"locate_ties__tol_{tol:g}"] = np.asarray(tree.locate_points(pts, tol))
        out["locate_ties__tolerances"] = np.array([1e-12, 1e-6, 1e-2])
    except TypeError:  # an older package without the tolerance argument
        out["locate_ties__default"] = np.asarray(tree.locate_points(pts))
        out["locate_ties__tolerances"] = np.array([])
    print("locate_ties:", pts.shape[0], "points")

    # ---- locate_points: the FORM of the on-edge tolerance.  A point at distance d outside a hull side of length L counts as
    # on the side if  |cross(side, point - start)| < tol * L  (d < tol: an absolute distance, what the oracle assumes,
    # oracle/xr_oracle.c:point_in_poly_or_on_edge)  or if  |cross| < tol  (d < tol / L: the threshold scales with 1 / L).
    # A long side (L = 100) and a short one (L = 0.01), probes at 0.9 tol and 1.1 tol (and far inside either threshold):
    #   long side,  d = 0.9 tol:  absolute form -> face 0,  cross form -> -1 (0.9 tol > tol / 100)
    #   short side, d = 1.1 tol:  absolute form -> -1,      cross form -> face 1 if the traversal reaches the face
    tol = 1.0e-3
    txy = np.array([[0.0, 0.0], [100.0, 0.0], [50.0, 40.0], [200.0, 0.0], [200.01, 0.0], [200.005, 0.008]])
    tfaces = np.array([[0, 1, 2], [3, 4, 5]], dtype=np.int64)
    d = np.array([0.005, 0.5, 0.9, 1.1, 2.0, 50.0, 150.0]) * tol
    tpts = np.vstack([np.column_stack([np.full(d.size, 50.0), -d]),          # below the middle of the long side
                      np.column_stack([np.full(d.size, 20.0), -d]),          # ... and off-centre
                      np.column_stack([np.full(d.size, 200.005), -d]),       # below the middle of the short side
                      [[50.0, 1.0], [200.005, 0.002]]])                      # interior points (sanity)
    ttree = CellTree2d(txy, tfaces, -1)
    out.update({"locate_tolform__xy": txy, "locate_tolform__faces": tfaces, "locate_tolform__points": tpts,
                "locate_tolform__tol": np.array(tol), "locate_tolform__d": d})
    try:
        out["locate_tolform__result"] = np.asarray(ttree.locate_points(tpts, tol))
    except TypeError:
        out["locate_tolform__result"] = np.asarray(ttree.locate_points(tpts))
        out["locate_tolform__tol"] = np.array(np.nan)
    print("locate_tolform:", tpts.shape[0], "points ->", out["locate_tolform__result"].tolist())

    # ---- barycentric weights: convex polygons and a concave one starting at its reflex corner
    bxy = np.array([
        [0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0],          # 0-3  unit square
        [2.0, 0.0], [2.0, 1.0],                                  # 4-5  square to the right
        [0.5, 0.6],                                              # 6    (unused node: the package must tolerate one)
        [3.0, 0.0], [4.0, 0.0], [4.5, 0.8], [3.5, 1.5], [2.8, 0.9],  # 7-11 convex pentagon
        [5.0, 0.0], [7.0, 0.0], [7.0, 2.0], [6.0, 0.5], [5.0, 2.0],  # 12-16 concave "M": reflex corner = node 15
    ])
    bfaces = np.array([
        [0, 1, 2, 3, -1],
        [1, 4, 5, 2, -1],
        [7, 8, 9, 10, 11],
        [15, 16, 12, 13, 14],        # the concave polygon, vertex list STARTING at the reflex corner (CCW overall)
    ], dtype=np.int64)
    rngb = np.random.default_rng(8)
    bpts = np.vstack([
        rngb.uniform([0, 0], [2, 1], (60, 2)), rngb.uniform([2.8, 0], [4.5, 1.5], (60, 2)), rngb.uniform([5, 0], [7, 2], (120, 2)),
        [[1.0, 0.5], [0.5, 0.0], [1.0, 1.0], [6.0, 0.5], [6.0, 0.25], [5.5, 1.0], [6.5, 1.0]],
    ])
    btree = CellTree2d(bxy, bfaces, -1)
    try:
        bi, bw = btree.compute_barycentric_weights(bpts, None)
    except TypeError:
        bi, bw = btree.compute_barycentric_weights(bpts)
    out.update({"bary_concave__xy": bxy, "bary_concave__faces": bfaces, "bary_concave__points": bpts,
                "bary_concave__face_index": np.asarray(bi), "bary_concave__weights": np.asarray(bw)})
    # the same concave polygon listed from a convex corner (no reversal expected)
    bfaces2 = bfaces.copy()
    bfaces2[3] = [12, 13, 14, 15, 16]
    btree2 = CellTree2d(bxy, bfaces2, -1)
    try:
        bi2, bw2 = btree2.compute_barycentric_weights(bpts, None)
    except TypeError:
        bi2, bw2 = btree2.compute_barycentric_weights(bpts)
    out.update({"bary_concave__faces_convex_start": bfaces2, "bary_concave__face_index_convex_start": np.asarray(bi2),
                "bary_concave__weights_convex_start": np.asarray(bw2)})
    print("bary_concave:", bpts.shape[0], "points")

    # ---- intersect_edges: corner touches, shared sides, ends on sides
    exy, ef = quads(4, 4)                                          # 4 x 4 unit-square cells of side 0.25
    edges = np.array([
        [[0.0, 0.0], [1.0, 1.0]],          # the diagonal: through every interior corner
        [[0.25, 0.0], [0.25, 1.0]],        # ALONG a shared side, full height
        [[0.0, 0.5], [1.0, 0.5]],          # along a horizontal shared side
        [[0.1, 0.1], [0.25, 0.1]],         # ends ON a side
        [[0.25, 0.1], [0.4, 0.1]],         # starts ON a side
        [[0.0, 0.25], [0.25, 0.0]],        # cuts a corner cell, passing through two nodes
        [[-0.5, 0.0], [1.5, 0.0]],         # outside along the hull
        [[0.5, 0.5], [0.5, 0.5]],          # zero-length, on a node
        [[0.1, 0.2], [0.9, 0.7]],          # general position
        [[0.9, 0.7], [0.1, 0.2]],          # ... and reversed
        [[1.0, 1.0], [2.0, 2.0]],          # touches the mesh in one corner only
        [[0.125, 0.125], [0.2, 0.15]],     # entirely inside one cell
    ])
    etree = CellTree2d(exy, ef, -1)
    ei, fi, seg = etree.intersect_edges(edges)
    order = np.lexsort((fi, ei))
    out.update({"edges_touch__xy": exy, "edges_touch__faces": ef, "edges_touch__edges": edges, "edges_touch__edge": np.asarray(ei)[order],
                "edges_touch__face": np.asarray(fi)[order], "edges_touch__segments": np.asarray(seg)[order]})
    # a triangle mesh and random segments
    redges = np.random.default_rng(4).uniform(-0.1, 1.1, (300, 2, 2))
    ttree = CellTree2d(p, f, -1)
    ei, fi, seg = ttree.intersect_edges(redges)
    order = np.lexsort((fi, ei))
    out.update({"edges_random__xy": p, "edges_random__faces": f, "edges_random__edges": redges, "edges_random__edge": np.asarray(ei)[order],
                "edges_random__face": np.asarray(fi)[order], "edges_random__segments": np.asarray(seg)[order]})
    print("edges:", edges.shape[0] + redges.shape[0], "segments")
    np.savez_compressed(path, **out)
    print("wrote", path, f"({os.path.getsize(path) / 1e6:.1f} MB) with numba_celltree", out["_numba_celltree_version"])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "g11_celltree.npz"))
