"""
CPU: the locate kernels' exact point-in-face test (xugrid_amd/csrc/xr_point_in_face.h) compiled as plain C++
(tests/host_point_in_face.cpp against tests/host_shim) and compared with the oracle's locate_points on one-face meshes.
The header only EVALUATES the oracle's square root and division where a cheap bound leaves their outcome open; this test
aims at exactly those places -- points on edges and vertices, at distance tol (1 +- a few ulp) of an edge, level with
vertices, within a few ulp of a crossing -- with a reciprocal degraded by up to 1e-6 relative, tolerances from 0 to 1e-3
and UTM-sized coordinates.  Every boolean must equal the oracle's.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("host_pif") / "host_point_in_face.so")
    subprocess.check_call(
        ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "tests", "host_shim"),
         "-I", os.path.join(ROOT, "xugrid_amd", "csrc"), os.path.join(ROOT, "tests", "host_point_in_face.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.host_point_in_face_many.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_double, ctypes.c_void_p]
    lib.host_filter_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_double, ctypes.c_void_p]
    lib.host_set_rcp_error.argtypes = [ctypes.c_double]
    return lib


def convex_polygon(rng, n, centre, radius):
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    while np.diff(np.concatenate([ang, [ang[0] + 2 * np.pi]])).max() > 0.9 * np.pi:  # keep it strictly convex and CCW
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    r = radius * rng.uniform(0.97, 1.0)
    return centre + r * np.column_stack([np.cos(ang), np.sin(ang)])


def nudge(x, k):
    for _ in range(abs(k)):
        x = np.nextafter(x, np.inf if k > 0 else -np.inf)
    return x


def adversarial_points(rng, poly, tol, n_random):
    n = poly.shape[0]
    lo, hi = poly.min(0), poly.max(0)
    span = (hi - lo).max()
    pts = [rng.uniform(lo - 0.2 * span, hi + 0.2 * span, (n_random, 2)), poly.copy()]
    for i in range(n):
        v0, v1 = poly[i - 1], poly[i]
        w = v1 - v0
        length = np.hypot(*w)
        normal = np.array([w[1], -w[0]]) / length  # outward for a CCW polygon
        t = rng.uniform(-0.05, 1.05, 24)[:, None]
        on = v0 + t * w
        pts.append(on)
        for scale in (1.0, 1.0 - 1e-15, 1.0 + 1e-15, 1.0 - 3e-8, 1.0 + 3e-8, 0.5, 2.0):
            pts.append(on + normal * tol * scale)
            pts.append(on - normal * tol * scale)
        # level with the end points, and within a few ulp of the crossing of the horizontal through p
        if w[1] != 0:
            py = rng.uniform(min(v0[1], v1[1]), max(v0[1], v1[1]), 12)
            py = np.concatenate([py, [v0[1], v1[1]]])
            xint = w[0] * (py - v0[1]) / w[1] + v0[0]
            for k in (-3, -1, 0, 1, 3):
                pts.append(np.column_stack([nudge(xint, k), py]))
            pts.append(np.column_stack([xint * (1 + 2e-6), py]))
            pts.append(np.column_stack([xint * (1 - 2e-6), py]))
    return np.ascontiguousarray(np.concatenate(pts))


def run_case(host, oracle, poly, pts, tol):
    n = poly.shape[0]
    faces = np.arange(n)[None, :]
    exp = oracle.CellTree2d(poly, faces).locate_points(pts, tol) >= 0
    poly_c = np.ascontiguousarray(poly)
    # the exact box test in front of the polygon test (oracle: locate_one; kernels: `consider` in xr_locate.hip)
    lo, hi = poly.min(0), poly.max(0)
    in_box = ~((pts[:, 0] < lo[0] - tol) | (pts[:, 0] > hi[0] + tol) | (pts[:, 1] < lo[1] - tol) | (pts[:, 1] > hi[1] + tol))
    for err in (0.0, 2e-7, -2e-7, 9e-7, -9e-7):
        host.host_set_rcp_error(err)
        got = np.empty(pts.shape[0], dtype=np.uint8)
        host.host_point_in_face_many(poly_c.ctypes.data, n, pts.ctypes.data, pts.shape[0], tol, got.ctypes.data)
        bad = np.nonzero((got.astype(bool) & in_box) != exp)[0]
        assert bad.size == 0, (err, tol, poly.tolist(), pts[bad[0]].tolist(), bool(got[bad[0]]), bool(exp[bad[0]]))
    host.host_set_rcp_error(0.0)
    return exp


def test_booleans_equal_the_oracle(host, oracle):
    rng = np.random.default_rng(5)
    n_inside = n_total = 0
    for case in range(240):
        n = int(rng.integers(3, 9))
        centre = np.array([0.0, 0.0]) if case % 3 else np.array([5.0e5, 6.0e6])
        radius = float(10.0 ** rng.uniform(-2, 3))
        poly = convex_polygon(rng, n, centre, radius)
        diag = np.hypot(*(poly.max(0) - poly.min(0)))
        tol = [0.0, 1e-12 * diag, 1e-3 * radius, 1e-300, 1e-6 * diag][case % 5]
        pts = adversarial_points(rng, poly, tol, 400)
        exp = run_case(host, oracle, poly, pts, tol)
        n_inside += int(exp.sum())
        n_total += exp.size
    assert n_total > 400_000 and 0.2 < n_inside / n_total < 0.8


def test_lattice_polygons_and_degenerate_edges(host, oracle):
    """Integer coordinates: points exactly on edges, on vertices, level with vertices; repeated vertices (zero-length edges are
    skipped by the test as by the oracle)."""
    rng = np.random.default_rng(6)
    gx, gy = np.meshgrid(np.arange(-1.0, 6.0, 0.5), np.arange(-1.0, 6.0, 0.5))
    pts = np.ascontiguousarray(np.column_stack([gx.ravel(), gy.ravel()]))
    polys = [np.array([[0.0, 0.0], [4.0, 0.0], [4.0, 4.0], [0.0, 4.0]]), np.array([[0.0, 0.0], [5.0, 1.0], [2.0, 4.0]]),
             np.array([[1.0, 0.0], [3.0, 0.0], [4.0, 2.0], [3.0, 4.0], [1.0, 4.0], [0.0, 2.0]]),
             np.array([[0.0, 0.0], [4.0, 0.0], [4.0, 0.0], [4.0, 4.0], [0.0, 4.0], [0.0, 4.0]])]
    for poly in polys:
        for shift in (np.zeros(2), np.array([5.0e5, 6.0e6])):
            for tol in (0.0, 1e-9, 0.25):
                run_case(host, oracle, poly + shift, np.ascontiguousarray(pts + shift), tol)
    del rng


def test_filters_decide_almost_everything(host):
    """On ordinary query points the square root is needed for < 1 % of the edges and the division for < 0.1 % of the crossings."""
    rng = np.random.default_rng(7)
    poly = np.ascontiguousarray(convex_polygon(rng, 6, np.array([5.0e5, 6.0e6]), 40.0))
    pts = np.ascontiguousarray(rng.uniform(poly.min(0) - 10, poly.max(0) + 10, (200_000, 2)))
    stats = np.zeros(4, dtype=np.int64)
    host.host_filter_stats(poly.ctypes.data, 6, pts.ctypes.data, pts.shape[0], 1e-12 * 120.0, stats.ctypes.data)
    assert stats[0] == 6 * 200_000 and stats[2] > 100_000
    assert stats[1] < 0.01 * stats[0] and stats[3] < 0.001 * stats[2]
