"""
CPU: the oracle (oracle/xr_oracle.c) against the golden vectors generated from the reference
(tests/golden/gen_goldens.py) and against the reference's own known answers, transcribed as
numbers.  This is what pins the oracle (DESIGN.md "Oracle").
"""
import numpy as np
import pytest

from conftest import same_or_nan
from xugrid_amd import meshgen


def method_arg(name):
    if name in ("p0", "p100") or "." in name:
        return ("percentile", float(name[1:]))
    return name


def test_g1_reducers_bit_exact(golden, oracle):
    g = golden("g1_reducers.npz")
    off, V, W = g["offsets"], g["values"], g["weights"]
    for key in g.files:
        if not key.startswith("out_"):
            continue
        name = key[4:]
        exp = g[key]
        got = np.array([oracle.reduce(method_arg(name), V[off[i]:off[i + 1]], W[off[i]:off[i + 1]]) for i in range(off.size - 1)])
        if name == "geometric_mean":  # libm exp/log vs numpy's: 1 ulp
            np.testing.assert_allclose(got, exp, rtol=1e-14, equal_nan=True)
        else:
            assert same_or_nan(got, exp).all(), name


def test_reduce_known_answers(oracle):
    # tests/test_regrid/test_reduce.py:7-79 (values [0,1,2,nan], weights 0.5)
    v = np.array([0.0, 1.0, 2.0, np.nan])
    w = np.full(4, 0.5)
    for vv in (v, v[::-1].copy()):
        assert oracle.reduce("mean", vv, w) == 1.0
        assert oracle.reduce("sum", vv, w) == 3.0
        assert oracle.reduce("minimum", vv, w) == 0.0
        assert oracle.reduce("maximum", vv, w) == 2.0
        assert oracle.reduce("median", vv, w) == 1.0
        assert oracle.reduce("first_order_conservative", vv, w) == 1.5
        assert oracle.reduce("max_overlap", vv, w) == 2.0
        assert oracle.reduce("mode", vv, w) == 2.0
        assert np.isclose(oracle.reduce("harmonic_mean", vv, w), 1.0 / (0.5 * (1.0 + 0.5)))
    # all-zero weights -> NaN, all-NaN values -> NaN (test_reduce.py:170-183)
    for m in ("mean", "harmonic_mean", "geometric_mean", "sum", "minimum", "maximum", "mode", "median",
              "first_order_conservative", "max_overlap", "p5", "p95"):
        assert np.isnan(oracle.reduce(m, v, np.zeros(4))), m
        assert np.isnan(oracle.reduce(m, np.full(4, np.nan), w)), m
    # create_percentile_method(50) on [0..4] -> 2 (test_regridder.py:282-293)
    assert oracle.reduce(("percentile", 50.0), np.arange(5.0), np.ones(5)) == 2.0


def test_g2_apply_bit_exact(golden, oracle):
    g = golden("g2_apply.npz")
    T = int(g["T"])
    for key in g.files:
        if not key.startswith("out64_"):
            continue
        name = key[6:]
        for tag in ("64", "32"):
            got = oracle.regrid_csr(method_arg(name), g["src" + tag], g["data"], g["indices"], g["indptr"], T)
            exp = g[f"out{tag}_{name}"]
            if name == "geometric_mean":
                np.testing.assert_allclose(got, exp, rtol=1e-14, equal_nan=True)
            else:
                assert same_or_nan(got, exp).all(), (name, tag)
            got2 = oracle.regrid_csr(method_arg(name), g["src" + tag], g["data"], g["indices"], g["indptr"], T, parallel_rows=True)
            assert same_or_nan(got, got2).all()
    got = oracle.regrid_coo(g["src64"], g["coo_row"], g["coo_col"], T)
    assert same_or_nan(got, g["coo_out64"]).all()


def test_g3_csr(golden, oracle):
    g = golden("g3_csr.npz")
    assert np.array_equal(oracle.to_csr_indptr(g["row"], int(g["n"])), g["indptr"])
    assert np.array_equal(oracle.to_csr_indptr(g["small_row"], 5), [0, 2, 4, 6, 8, 10])  # tests/test_sparse.py:37-43


def test_g5_area_centroids_bit_exact(golden, oracle):
    g = golden("g5_geometry.npz")
    for t in ("tri", "quad", "mix"):
        assert np.array_equal(oracle.area(g[t + "_xy"], g[t + "_faces"]), g[t + "_area"])
        assert np.array_equal(oracle.centroids(g[t + "_xy"], g[t + "_faces"]), g[t + "_centroids"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g4_rectilinear_overlap(golden, oracle, tag):
    """quad x quad polygon clip == separable overlap_1d x broadcast of the reference; this pins
    the unstructured clip against the rectilinear path as tests/test_regrid/test_regridder.py:295-332 does."""
    g = golden("g4_rectilinear.npz")
    sxy, sf = meshgen.quad_mesh(g[tag + "_xe_s"], g[tag + "_ye_s"])
    txy, tf = meshgen.quad_mesh(g[tag + "_xe_t"], g[tag + "_ye_t"])
    tree = oracle.CellTree2d(sxy, sf)
    q, s, a = tree.intersect_faces(txy, tf)
    assert np.array_equal(q, g[tag + "_tgt"]) and np.array_equal(s, g[tag + "_src"])
    np.testing.assert_allclose(a, g[tag + "_w"], rtol=1e-12)
    if tag == "a":  # 36 pairs of 625 m2 (tests/test_regrid/test_structured.py:108-204)
        assert a.size == 36 and np.all(a == 625.0)
    # SAT pre-filter (as the reference) and brute force give the identical result
    q2, s2, a2 = tree.intersect_faces(txy, tf, use_sat=False)
    assert np.array_equal(q, q2) and np.array_equal(s, s2) and np.array_equal(a, a2)
    q3, s3, a3 = tree.intersect_faces_bruteforce(txy, tf)
    assert np.array_equal(q, q3) and np.array_equal(s, s3) and np.array_equal(a, a3)


def grid2d():
    """The 2-quad + 2-triangle mesh of tests/test_ugrid2d.py (numbers transcribed)."""
    xy = np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0], [0.0, 1.0], [1.0, 1.0], [2.0, 1.0], [1.0, 2.0]])
    faces = np.array([[0, 1, 4, 3], [1, 2, 5, 4], [3, 4, 6, -1], [4, 5, 6, -1]])
    return xy, faces


def test_locate_points_known_answers(oracle):
    xy, faces = grid2d()
    tree = oracle.CellTree2d(xy, faces)
    cen = oracle.centroids(xy, faces)
    assert np.array_equal(tree.locate_points(cen), [0, 1, 2, 3])
    # tests/test_ugrid2d.py:724-730: 0.01 outside, found with tolerance 0.011
    off = np.array([[-0.01, 1.0], [-0.01, 0.5]])
    assert np.array_equal(tree.locate_points(off, 0.011), [0, 0])
    assert np.array_equal(tree.locate_points(off), [-1, -1])
    # the point selections of the reference's own tests, which are locate_points results (ugridbase.py:1323):
    # tests/test_ugrid2d.py:835-859 (sel_points), :862-885 (out of bounds -> -1), :1085-1108 (sel on an outer product of x and y)
    assert np.array_equal(tree.locate_points(np.array([[0.5, 0.5], [1.5, 1.25]])), [0, 3])
    oob = np.array([[-10.0, -10.0], [0.5, 0.5], [-20.0, -20.0], [1.5, 1.25], [-30.0, -30.0]])
    assert np.array_equal(tree.locate_points(oob), [-1, 0, -1, 3, -1])
    gx, gy = np.meshgrid([0.4, 0.8, 1.2], [0.5, 1.1])
    assert np.array_equal(tree.locate_points(np.column_stack([gx.ravel(), gy.ravel()])), [0, 0, 1, 2, 2, 3])


def test_barycentric_known_answers(oracle):
    # tests/test_ugrid2d.py:751-791
    xy, faces = grid2d()
    tree = oracle.CellTree2d(xy, faces)
    pts = np.array([[0.0, 0.0], [0.5, 0.5], [1.5, 0.5], [0.5, 1.5], [2.0, 2.0]])
    face, w = tree.compute_barycentric_weights(pts)
    assert np.array_equal(face, [0, 0, 1, 2, -1])
    expected = np.array([[1.0, 0, 0, 0], [0.25] * 4, [0.25] * 4, [0.5, 0.0, 0.5, 0.0], [0.0] * 4])
    np.testing.assert_allclose(w, expected, atol=1e-15)
    pts2 = pts.copy()
    pts2[:, 0] -= 0.01
    face, w = tree.compute_barycentric_weights(pts2, tolerance=0.01)
    assert np.array_equal(face, [-1, 0, 1, 2, -1])
    expected[0] = 0.0
    np.testing.assert_allclose(w, expected, atol=0.05)


def test_barycentric_concave_reference_known_answer(oracle):
    """The reference's test_barycentric_concave (tests/test_regrid/test_regridder.py:334-369), numbers transcribed: three
    triangles around a reflex corner -> 30 x 20 raster of 0.1-wide cells; exactly 200 cells stay NaN (the gap right of the
    reflex corner), every value lies in [0.5, 2.0].  One of the few reference tests that pin the external locate +
    barycentric arithmetic.  Oracle step by step: unstructured.py:146-201, then the mean apply."""
    import xugrid_amd as xa
    from stepwise import CONCAVE_FACES, CONCAVE_VALUES, CONCAVE_VERTICES, concave_raster_axes, oracle_barycentric_triplets

    grid = xa.Ugrid2d(CONCAVE_VERTICES[:, 0], CONCAVE_VERTICES[:, 1], -1, CONCAVE_FACES)
    x, y = concave_raster_axes()
    yy, xx = np.meshgrid(y, x, indexing="ij")
    points = np.column_stack([xx.ravel(), yy.ravel()])
    s, t, w = oracle_barycentric_triplets(oracle, grid, points)
    out = oracle.regrid_csr("mean", CONCAVE_VALUES[None, :], w, s, oracle.to_csr_indptr(t, points.shape[0]), points.shape[0])[0]
    out = out.reshape(y.size, x.size)
    assert np.isnan(out).sum() == 200
    assert np.nanmin(out) >= 0.5 and np.nanmax(out) <= 2.0
    # the gap is the wedge between faces 0 and 2 right of the reflex corner (1, 1): nothing left of x = 1 is NaN
    assert not np.isnan(out[:, x < 1.0]).any() and np.isnan(out[:, x > 1.0]).sum() == 200


def test_self_overlap_identity(oracle):
    """tests/test_regrid/test_unstructured.py:32-45 on a seeded triangle mesh."""
    xy, faces = meshgen.triangle_mesh(400, 7)
    tree = oracle.CellTree2d(xy, faces)
    q, s, a = tree.intersect_faces(xy, faces)
    valid = a > 1e-5 * a.max()
    assert np.array_equal(q[valid], np.arange(faces.shape[0])) and np.array_equal(s[valid], q[valid])
    np.testing.assert_allclose(a[valid], oracle.area(xy, faces), rtol=1e-12)
    # locate_centroids / barycentric identities (:47-59)
    cen = oracle.centroids(xy, faces)
    assert np.array_equal(tree.locate_points(cen), np.arange(faces.shape[0]))


def test_clip_against_exact_rational(oracle):
    """Sutherland-Hodgman area vs an exact rational clip (oracle/exact_clip.py) on random convex polygons."""
    from oracle import exact_clip

    rng = np.random.default_rng(11)
    worst = 0.0
    for _ in range(300):
        a = exact_clip.random_convex(rng, int(rng.integers(3, 7)))
        b = exact_clip.random_convex(rng, int(rng.integers(3, 7))) + rng.uniform(-0.5, 0.5, 2)
        exact = float(exact_clip.intersection_area(a, b))
        got = oracle.clip_area(a, b)
        scale = min(float(exact_clip.polygon_area(a)), float(exact_clip.polygon_area(b)))
        worst = max(worst, abs(got - exact) / scale)
    assert worst < 1e-12


def test_conservation_elevation_nl(golden, oracle):
    """Config 1 plumbing on the CPU oracle: total overlap area == mesh area."""
    g = golden("g8_elevation_nl.npz")
    xy = np.column_stack([g["node_x"], g["node_y"]])
    faces = g["face_nodes"].astype(np.int64)
    xmin, ymin, xmax, ymax = xy[:, 0].min(), xy[:, 1].min(), xy[:, 0].max(), xy[:, 1].max()
    txy, tf = meshgen.quad_mesh(np.linspace(xmin, xmax, 201), np.linspace(ymin, ymax, 201))
    tree = oracle.CellTree2d(xy, faces)
    q, s, a = tree.intersect_faces(txy, tf)
    total = oracle.area(xy, faces).sum()
    assert abs(a.sum() / total - 1) < 1e-10
    assert abs(total / 4.2169478944e10 - 1) < 1e-9
    cell_area = oracle.area(txy, tf)
    row_sum = np.bincount(q, weights=a, minlength=tf.shape[0])
    assert (row_sum <= cell_area * (1 + 1e-10)).all()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g10_replace_interpolated_weights(golden, oracle, tag):
    """G10: replace_interpolated_weights (xugrid/regrid/unstructured.py:17-57) lifted from the reference
    (tests/golden/gen_replace.py): the oracle's restatement and the product's host-side counterpart reproduce the
    weights the reference leaves behind bit for bit -- hot-path-like cells, repeated q / r / p in a face, chains."""
    from xugrid_amd._replace import replace_interpolated_weights as host_replace

    g = golden("g10_replace.npz")
    args = [g[f"{tag}_{k}"] for k in ("vertices", "faces", "face_index")]
    node_map, threshold, exp = g[f"{tag}_node_to_node_map"], int(g[f"{tag}_threshold"]), g[f"{tag}_weights_out"]
    assert (exp != g[f"{tag}_weights_in"]).sum() > 5000  # the case does exercise the function
    w = g[f"{tag}_weights_in"].copy()
    oracle.replace_interpolated_weights(*args, w, node_map, threshold)
    assert np.array_equal(w, exp)
    w = g[f"{tag}_weights_in"].copy()
    host_replace(*args, w, node_map, threshold)
    assert np.array_equal(w, exp)


def test_oracle_runs_clean_under_sanitizers():
    """the C oracle under AddressSanitizer + UndefinedBehaviorSanitizer on small seeded inputs --
    every entry point of xr_oracle.h (tree, overlap incl. SAT and brute force, every reducer, COO, area / centroids,
    locate with and without tolerance, barycentric weights, network edges); any finding aborts the driver."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "sanitize"], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    assert "sanitize_driver ok" in proc.stdout
