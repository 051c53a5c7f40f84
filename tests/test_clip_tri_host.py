"""
CPU: the device-side triangle x triangle clip bookkeeping (xugrid_amd/csrc/xr_clip_tri.h: inside-flag bit masks,
compaction table, per-lane generic fallback) compiled as plain C++ for one "lane" (tests/host_clip_tri.cpp against
tests/host_shim/hip/hip_runtime.h) and compared BIT FOR BIT with the oracle's clip_polygons -- the restatement of
numba_celltree's Sutherland-Hodgman clip -- on random, nearly coincident, lattice-aligned (exactly degenerate),
shared-vertex / shared-edge and repeated-vertex triangle pairs.  The GPU tests check the same function in its
kernel; this test exercises the rare branches (repeated vertices, parallel crossing edges, > 2 transitions) densely.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_clip(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("host_clip") / "host_clip_tri.so")
    subprocess.check_call(
        ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "tests", "host_shim"),
         "-I", os.path.join(ROOT, "xugrid_amd", "csrc"), os.path.join(ROOT, "tests", "host_clip_tri.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.host_tri_clip_many.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
    lib.host_quad_clip_many.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]

    def many(tv, sv):
        tv = np.ascontiguousarray(tv, dtype=np.float64)
        sv = np.ascontiguousarray(sv, dtype=np.float64)
        out = np.empty(tv.shape[0])
        lib.host_tri_clip_many(tv.ctypes.data, sv.ctypes.data, tv.shape[0], out.ctypes.data)
        return out

    def quads(tv, n0, sv):
        tv = np.ascontiguousarray(tv, dtype=np.float64)
        sv = np.ascontiguousarray(sv, dtype=np.float64)
        n0 = np.ascontiguousarray(n0, dtype=np.int32)
        out = np.empty(tv.shape[0])
        lib.host_quad_clip_many(tv.ctypes.data, n0.ctypes.data, sv.ctypes.data, tv.shape[0], out.ctypes.data)
        return out

    many.quads = quads
    return many


def ccw(t):
    u, v = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]
    cw = (u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]) < 0
    t[cw] = t[cw][:, ::-1]
    return t


def check(host_clip, oracle, tv, sv, min_positive):
    got = host_clip(tv, sv)
    exp = np.array([oracle.clip_area(tv[i], sv[i]) for i in range(tv.shape[0])])
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, (bad.size, tv[bad[0]].tolist(), sv[bad[0]].tolist(), got[bad[0]], exp[bad[0]])
    assert (exp > 0).sum() >= min_positive


def test_random_and_near_pairs(host_clip, oracle):
    rng = np.random.default_rng(0)
    n = 60_000
    tv, sv = ccw(rng.random((n, 3, 2))), ccw(rng.random((n, 3, 2)) * 0.8 + 0.1)
    c = rng.random((n, 1, 2))
    tv2, sv2 = ccw(c + 0.2 * rng.random((n, 3, 2))), ccw(c + 0.2 * rng.random((n, 3, 2)))
    check(host_clip, oracle, np.concatenate([tv, tv2]), np.concatenate([sv, sv2]), 50_000)


def test_lattice_aligned_pairs(host_clip, oracle):
    """Integer coordinates on a 5 x 5 lattice: vertices on clip lines, collinear and repeated vertices, shared
    edges, identical triangles -- every cross product that is zero is exactly zero."""
    rng = np.random.default_rng(1)
    n = 150_000
    tv = rng.integers(0, 5, size=(n, 3, 2)).astype(np.float64)
    sv = rng.integers(0, 5, size=(n, 3, 2)).astype(np.float64)
    tv, sv = ccw(tv), ccw(sv)
    sv[::7] = tv[::7]  # identical
    sv[1::7, :2] = tv[1::7, 1::-1]  # shared edge, opposite direction
    check(host_clip, oracle, tv, sv, 20_000)
    # UTM-like offsets keep the exact degeneracies
    check(host_clip, oracle, tv[:30_000] * 25.0 + np.array([5.0e5, 6.0e6]), sv[:30_000] * 25.0 + np.array([5.0e5, 6.0e6]), 4_000)


def test_shared_vertex_mesh_pairs(host_clip, oracle):
    """Neighbouring triangles of one Delaunay mesh and of its refinement (a mesh against itself): vertices ON the
    clip lines, crossing points that coincide with vertices."""
    from xugrid_amd import meshgen

    xy, faces = meshgen.triangle_mesh(400, 3)
    tri = xy[faces]
    cen = tri.mean(axis=1)
    d = ((cen[:, None, :] - cen[None, :, :]) ** 2).sum(axis=2)
    near = np.argsort(d, axis=1)[:, :14]
    i = np.repeat(np.arange(faces.shape[0]), near.shape[1])
    j = near.ravel()
    check(host_clip, oracle, tri[i], tri[j], faces.shape[0])
    # refinement: every triangle split at its edge midpoints, against the coarse neighbours
    mid = 0.5 * (tri + np.roll(tri, -1, axis=1))
    fine = np.concatenate([np.stack([tri[:, k], mid[:, k], mid[:, k - 1]], axis=1) for k in range(3)] + [mid])
    fi = np.repeat(np.arange(fine.shape[0]), 6)
    owner = np.tile(np.arange(faces.shape[0]), 4)
    fj = near[owner][:, :6].ravel()
    check(host_clip, oracle, ccw(fine[fi].copy()), tri[fj], fine.shape[0])


def check_quads(host_clip, oracle, tv, n0, sv, min_positive):
    got = host_clip.quads(tv, n0, sv)
    exp = np.array([oracle.clip_area(tv[i, : n0[i]], sv[i]) for i in range(tv.shape[0])])
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, (bad.size, tv[bad[0]].tolist(), int(n0[bad[0]]), sv[bad[0]].tolist(), got[bad[0]], exp[bad[0]])
    assert (exp > 0).sum() >= min_positive


def convex_quads(rng, n, centre, size):
    ang = np.sort(rng.uniform(0, 2 * np.pi, (n, 4)), axis=1)
    return centre + size * np.stack([np.cos(ang), np.sin(ang)], axis=2)


def test_quadrilateral_subjects(host_clip, oracle):
    """The MAXV = 7 instantiation (faces of a raster / quadrilateral target against source triangles): random convex quads,
    axis-aligned cells against a triangle mesh's own triangles (vertices ON cell lines: lattice coordinates), cells that are
    triangles with a fill slot (n0 = 3), repeated vertices, UTM-sized offsets."""
    rng = np.random.default_rng(4)
    n = 60_000
    c = rng.random((n, 1, 2))
    tv = convex_quads(rng, n, c, 0.15)
    sv = ccw(c + 0.3 * (rng.random((n, 3, 2)) - 0.5))
    check_quads(host_clip, oracle, tv, np.full(n, 4), sv, 30_000)
    # raster cells on an integer lattice against lattice triangles: every degenerate contact is exact
    n = 120_000
    x0 = rng.integers(0, 4, (n, 1)).astype(np.float64); y0 = rng.integers(0, 4, (n, 1)).astype(np.float64)
    w = rng.integers(1, 3, (n, 1)).astype(np.float64); h = rng.integers(1, 3, (n, 1)).astype(np.float64)
    cells = np.stack([np.hstack([x0, y0]), np.hstack([x0 + w, y0]), np.hstack([x0 + w, y0 + h]), np.hstack([x0, y0 + h])], axis=1)
    tris = ccw(rng.integers(0, 6, size=(n, 3, 2)).astype(np.float64))
    n0 = np.full(n, 4)
    check_quads(host_clip, oracle, cells, n0, tris, 20_000)
    check_quads(host_clip, oracle, cells[:30_000] * 25.0 + np.array([5.0e5, 6.0e6]), n0[:30_000], tris[:30_000] * 25.0 + np.array([5.0e5, 6.0e6]), 4_000)
    # half of the "cells" are triangles (fill slot): the fourth vertex is never read
    tri_cells = cells.copy()
    n0 = np.where(rng.random(n) < 0.5, 3, 4)
    tri_cells[n0 == 3, 3] = np.nan
    check_quads(host_clip, oracle, tri_cells, n0, tris, 15_000)
    # repeated vertices (a quad with a zero-length side) and general lattice quads, not all convex-regular
    rep = cells.copy()
    rep[::3, 2] = rep[::3, 1]
    check_quads(host_clip, oracle, rep, np.full(n, 4), tris, 10_000)
