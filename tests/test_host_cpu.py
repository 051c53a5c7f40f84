"""
CPU: host-side logic that needs no GPU -- the C-ABI library loads and exports every symbol the
header declares, the product never touches oracle/, sparse containers, connectivity helpers, the
Voronoi pre-step against goldens G6, raster -> quad conversion, error behaviour.
"""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    from xugrid_amd import _lib

    header = open(os.path.join(ROOT, "include", "xugrid_amd.h")).read()
    declared = set(re.findall(r"\b(xr_[a-z0-9_]+)\s*\(", header))
    declared -= {"xr_last_error"} - set(re.findall(r"\*(xr_last_error)\(", header))
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing
    # ... and every declared symbol is bound in the ctypes table (and nothing else)
    assert set(_lib.SIGNATURES) == declared
    assert _lib.load().xr_version() >= 100


def test_no_silent_cpu_fallback():
    """Without a device every compute entry point must raise; device_count itself must not."""
    from xugrid_amd import _lib, engine

    if _lib.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_lib.XugridAmdError):
        engine.DeviceMesh(np.zeros((3, 2)), np.array([[0, 1, 2]]))
    with pytest.raises(_lib.XugridAmdError):
        engine.init(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "xugrid_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, re.M) or "xr_oracle" in text.replace(
                    "oracle/xr_oracle.c", ""
                ) or "libxr_oracle" in text:
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
    # importing the product must not load the oracle module
    code = "import sys; sys.path.insert(0, %r); import xugrid_amd; assert not any(m.startswith('oracle') for m in sys.modules), 'oracle imported'" % ROOT
    subprocess.check_call([sys.executable, "-c", code])


def test_sparse_containers(golden):
    from xugrid_amd.sparse import MatrixCOO, MatrixCSR

    g = golden("g3_csr.npz")
    A = MatrixCSR.from_triplet(g["row"], g["col"], g["data"], n=int(g["n"]), m=int(g["m"]))
    assert np.array_equal(A.indptr, g["indptr"]) and np.array_equal(A.indices, g["indices"])
    B = A.to_coo()
    assert np.array_equal(B.row, g["coo_row"]) and np.array_equal(B.col, g["coo_col"])
    small = MatrixCSR.from_triplet(g["small_row"], np.arange(10) % 3, np.ones(10))
    assert np.array_equal(small.indptr, [0, 2, 4, 6, 8, 10]) and small.n == 5 and small.m == 3
    C = MatrixCOO.from_triplet(np.array([0, 1]), np.array([3, 1]), np.ones(2), n=7, m=9)  # shape override
    assert (C.n, C.m, C.nnz) == (7, 9, 2)
    # row helpers (core/sparse.py:129-158)
    from xugrid_amd.sparse import columns_and_values, nzrange, row_slice

    for r in (0, 3, A.n - 1):
        sl = row_slice(A, r)
        assert list(nzrange(A, r)) == list(range(sl.start, sl.stop))
        assert [(c, v) for c, v in columns_and_values(A, sl)] == list(zip(A.indices[sl], A.data[sl]))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_voronoi_topology_matches_reference(golden, tag):
    from xugrid_amd import connectivity as C, voronoi

    g = golden("g6_voronoi.npz")
    xy, faces, cen = g[tag + "_xy"], g[tag + "_faces"], g[tag + "_centroids"]
    enc, fec = C.edge_connectivity(faces)
    efc = C.invert_dense(fec)
    assert np.array_equal(enc, g[tag + "_enc"]) and np.array_equal(efc, g[tag + "_efc"])
    nfc = C.invert_dense_to_sparse(faces, n_rows=len(xy))
    v, f, fi, nm = voronoi.voronoi_topology(nfc, xy, cen, efc, enc, add_exterior=True, add_vertices=True, skip_concave=True)
    assert np.array_equal(v, g[tag + "_vor_vertices"])
    assert np.array_equal(f, g[tag + "_vor_faces"])
    assert np.array_equal(fi, g[tag + "_vor_face_i"])
    # the reference pairs the two projections with a non-stable argsort: compare rows as sets
    assert np.array_equal(np.sort(nm, axis=1), np.sort(g[tag + "_vor_nmap"], axis=1))


def test_voronoi_requires_edges_for_exterior():
    from xugrid_amd import connectivity as C, voronoi

    faces = np.array([[0, 1, 2]])
    nfc = C.invert_dense_to_sparse(faces)
    with pytest.raises(ValueError):
        voronoi.voronoi_topology(nfc, np.zeros((3, 2)), np.zeros((1, 2)), add_exterior=True)


def test_raster_to_quads_orientation_and_numbering():
    """from_structured_bounds: face id = row-major (y, x) in the raster's own order; quads CCW
    for every axis direction combination (incl. the nx > ny descending-y case)."""
    from xugrid_amd.regrid.structured import Raster, StructuredGrid2d
    from xugrid_amd.regrid.unstructured import UnstructuredGrid2d

    for x in (np.array([25.0, 75.0, 125.0, 175.0, 225.0]), np.array([225.0, 175.0, 125.0, 75.0, 25.0])):
        for y in (np.array([175.0, 125.0, 75.0]), np.array([75.0, 125.0, 175.0])):
            grid = StructuredGrid2d(Raster(x, y, dx=50.0, dy=-50.0))
            assert grid.shape == (3, 5) and grid.size == 15 and grid.dims == ("y", "x")
            ug = grid.convert_to(UnstructuredGrid2d).ugrid_topology
            xy = ug.node_coordinates
            f = ug.face_node_connectivity
            p = xy[f]
            cx, cy = p[:, :, 0].mean(axis=1), p[:, :, 1].mean(axis=1)
            yy, xx = np.meshgrid(y, x, indexing="ij")
            assert np.allclose(cx, xx.ravel()) and np.allclose(cy, yy.ravel())
            u, v = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
            assert ((u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]) > 0).all()
    with pytest.raises(ValueError):
        StructuredGrid2d(Raster(np.array([0.0, 1.0, 3.0]), np.array([0.0, 1.0])))  # not equidistant
    with pytest.raises(ValueError):
        StructuredGrid2d(Raster(np.array([0.0, 2.0, 1.0]), np.array([0.0, 1.0])))  # not monotonic


def test_method_tables_and_errors():
    from xugrid_amd import reduce as R

    assert set(R.ABSOLUTE_OVERLAP_METHODS) == {
        "mean", "harmonic_mean", "geometric_mean", "sum", "minimum", "maximum", "mode", "median",
        "max_overlap", "p5", "p10", "p25", "p50", "p75", "p90", "p95",
    }
    assert set(R.RELATIVE_OVERLAP_METHODS) == {"conductance", "first_order_conservative"}
    assert R.create_percentile_method(33.3).percentile == 33.3
    with pytest.raises(ValueError):
        R.create_percentile_method(101.0)
    with pytest.raises(ValueError):
        R.create_percentile_method(-1.0)


def test_meshgen_is_deterministic_and_ccw():
    from xugrid_amd import meshgen

    for delaunay in (True, False):
        xy, f = meshgen.triangle_mesh(500, 3, 30.0, 0.7, delaunay=delaunay)
        xy2, f2 = meshgen.triangle_mesh(500, 3, 30.0, 0.7, delaunay=delaunay)
        assert np.array_equal(xy, xy2) and np.array_equal(f, f2)
        p = xy[f]
        u, v = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
        assert ((u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]) > 0).all()


def test_replace_interpolated_weights_host_equals_oracle(oracle):
    """unstructured.py:17-57: the host restatement (numpy) and the oracle (C) on 40 000 random rows, bit for bit
    (the host version once used ``** 2`` = libm pow, which is not always the correctly rounded square)."""
    from xugrid_amd._replace import replace_interpolated_weights

    rng = np.random.default_rng(21)
    n_vertex, n_extra, m, n_face, n = 400, 60, 7, 300, 40_000
    vertices = rng.random((n_vertex, 2)) * 10.0 ** rng.integers(-3, 4)
    threshold = n_vertex - n_extra
    faces = rng.integers(0, n_vertex, (n_face, m)).astype(np.int64)
    faces[rng.random(faces.shape) < 0.15] = -1
    n2n = rng.integers(0, threshold, (n_extra, 2)).astype(np.int64)
    # make the mapped neighbours appear in many faces
    for f in range(n_face):
        for j in range(m):
            if faces[f, j] >= threshold and rng.random() < 0.8:
                q, r = n2n[faces[f, j] - threshold]
                faces[f, (j + 1) % m], faces[f, (j + 2) % m] = q, r
    face_index = rng.integers(-1, n_face, n).astype(np.int64)
    weights = rng.random((n, m))
    weights[rng.random(weights.shape) < 0.2] = 0.0
    a, b = weights.copy(), weights.copy()
    replace_interpolated_weights(vertices=vertices, faces=faces, face_index=face_index, weights=a, node_to_node_map=n2n,
                                 node_index_threshold=threshold)
    oracle.replace_interpolated_weights(vertices, faces, face_index, b, n2n, threshold)
    assert not np.array_equal(a, weights)  # something was redistributed
    assert np.array_equal(a, b)


def test_rectilinear_ugrid_host_side_is_lazy_and_equal():
    """RectilinearUgrid2d (quads generated on the device) exposes the same host arrays as
    Ugrid2d.from_structured_bounds, but only materialises them when they are read."""
    import xugrid_amd as xa
    from xugrid_amd.ugrid2d import RectilinearUgrid2d

    rng = np.random.default_rng(2)
    for flip_x in (False, True):
        for flip_y in (False, True):
            xv = np.concatenate(([1.0], 1.0 + np.cumsum(rng.uniform(0.5, 2.0, 11))))
            yv = np.concatenate(([-2.0], -2.0 + np.cumsum(rng.uniform(0.5, 2.0, 7))))
            xv, yv = (xv[::-1].copy() if flip_x else xv), (yv[::-1].copy() if flip_y else yv)
            xb = np.column_stack([np.minimum(xv[:-1], xv[1:]), np.maximum(xv[:-1], xv[1:])])
            yb = np.column_stack([np.minimum(yv[:-1], yv[1:]), np.maximum(yv[:-1], yv[1:])])
            host = xa.Ugrid2d.from_structured_bounds(xb, yb)
            lazy = xa.Ugrid2d.from_structured_bounds_device(xb, yb)
            assert isinstance(lazy, RectilinearUgrid2d) and lazy._host is None
            assert (lazy.n_node, lazy.n_face, lazy.n_max_node_per_face) == (host.n_node, host.n_face, 4)
            assert lazy.bounds == host.bounds
            nodes = rng.integers(0, host.n_node, 20)
            assert np.array_equal(lazy.node_coordinates_of(nodes), host.node_coordinates_of(nodes))
            assert lazy._host is None  # sizes, bounds and single node coordinates need no host arrays
            assert np.array_equal(lazy.face_node_connectivity, host.face_node_connectivity)
            assert np.array_equal(lazy.node_x, host.node_x) and np.array_equal(lazy.node_y, host.node_y)
            again = xa.Ugrid2d.from_dataset(lazy.to_dataset("g"), "g")
            assert np.array_equal(again.face_node_connectivity, host.face_node_connectivity)
    with pytest.raises(ValueError):
        RectilinearUgrid2d(np.array([0.0]), np.array([0.0, 1.0]))


def test_run_time_options_are_an_api_not_the_environment():
    """The library's switches are options (xr_set_option / xr_get_option; read once from XR_<NAME> when the library is first
    used): known names round-trip, unknown names are an error, and no compute path looks at the environment (round-5 review:
    ~60 getenv reads, some per call, from threads that race with setenv)."""
    import glob
    import re

    from xugrid_amd import engine

    assert engine.get_option("overlap_fused") == 1 and engine.get_option("apply_contract") == 0
    assert engine.set_option("plan_merge", 1) == -1 and engine.get_option("plan_merge") == 1
    with engine.option("plan_merge", 0):
        assert engine.get_option("plan_merge") == 0
    assert engine.get_option("plan_merge") == 1
    engine.set_option("plan_merge", -1)
    with pytest.raises(ValueError):
        engine.set_option("no_such_option", 1)
    with pytest.raises(ValueError):
        engine.get_option("XR_OVERLAP_FUSED")
    # ONE getenv call site in the device library's sources
    calls = []
    for path in glob.glob(os.path.join(ROOT, "xugrid_amd", "csrc", "*.h*")):
        text = re.sub(r"//[^\n]*", "", open(path).read())
        calls += [path for _ in re.finditer(r"\bgetenv\s*\(", text)]
    assert len(calls) == 1 and calls[0].endswith("xr_engine.hip"), calls


def test_ugrid2d_coordinates_cannot_go_stale():
    """``node_x`` / ``node_y`` are plain attributes in the reference (ugrid2d.py:86-87) and ``node_coordinates`` a fresh array per
    call (ugridbase.py:576-579).  Here the interleaved table feeds the device: ``node_coordinates`` is a copy as in the reference (an
    in-place write changes nothing of the grid) and assigning an axis rebuilds the table and drops what was derived."""
    import xugrid_amd as xa

    xy, faces = xa.meshgen.triangle_mesh(100, 0)
    g = xa.Ugrid2d(xy[:, 0], xy[:, 1], -1, faces)
    nc = g.node_coordinates
    assert np.array_equal(nc, xy)
    nc[0, 0] = 5.0  # (a fresh array, writable as the reference's: the grid does not see it)
    assert np.array_equal(g.node_coordinates, xy) and g.node_x[0] == xy[0, 0]
    old_x = g.node_x.copy()
    g._area = np.ones(3)  # (stand-ins for device results: no device in this test)
    g._celltree = object()
    g.node_x = old_x * 2.0
    assert np.array_equal(g.node_x, old_x * 2.0) and np.array_equal(g.node_coordinates[:, 0], old_x * 2.0)
    assert np.array_equal(g.node_coordinates[:, 1], xy[:, 1]) and np.array_equal(g.node_y, xy[:, 1])
    assert g._area is None and g._celltree is None
    g.node_y = xy[:, 1] + 1.0
    assert np.array_equal(g.node_coordinates[:, 1], xy[:, 1] + 1.0)
    with pytest.raises(ValueError):
        g.node_x = np.zeros(3)


def test_custom_reduction_callable_on_cached_weights():
    """A caller's own reduction (regridder.py:136-137; examples/overlap_regridder.py:105-169) over cached weights: the loop of
    make_regrid (regridder.py:41-67) -- NaN-initialised output, empty rows untouched, values in the order of the row's indices,
    a workspace of the row's length -- needs no device."""
    import xugrid_amd as xa

    xy = np.array([[0.0, 0], [1, 0], [2, 0], [3, 0], [0, 1], [1, 1], [2, 1], [3, 1]])
    quads = np.array([[0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6]])
    src = xa.Ugrid2d(xy[:, 0], xy[:, 1], -1, quads)
    txy = np.array([[0.0, 0], [1.5, 0], [3, 0], [0, 1], [1.5, 1], [3, 1], [4, 0], [4, 1]])
    tgt = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, np.array([[0, 1, 4, 3], [1, 2, 5, 4], [2, 6, 7, 5]]))
    ds = {"__regrid_data": np.array([1.0, 0.5, 0.5, 1.0]), "__regrid_indices": np.array([0, 1, 1, 2]),
          "__regrid_indptr": np.array([0, 2, 4, 4]), "__regrid_n": np.array(3), "__regrid_m": np.array(3), "__regrid_nnz": np.array(4)}
    ds.update(xa.regrid.UnstructuredGrid2d(src).to_dataset("__source"))
    seen = []

    def own(values, weights, workspace):
        seen.append((values.copy(), weights.copy(), workspace.shape))
        workspace[:] = values * weights
        return workspace.sum() if not np.isnan(values).all() else np.nan

    data = np.array([[1.0, 2.0, 4.0], [np.nan, 2.0, 4.0]])
    out = xa.OverlapRegridder.from_weights(ds, tgt, method=own).regrid(data)
    assert out.shape == (2, 3)
    assert np.array_equal(out[0], [2.0, 5.0, np.nan], equal_nan=True)  # (row 2 is empty: stays NaN, the callable is not called)
    assert np.isnan(out[1, 0]) and out[1, 1] == 5.0 and np.isnan(out[1, 2])
    assert len(seen) == 4 and np.array_equal(seen[1][0], [2.0, 4.0]) and np.array_equal(seen[1][1], [0.5, 1.0]) and seen[1][2] == (2,)
    assert xa.OverlapRegridder.from_weights(ds, tgt, method=lambda v, w, ws: float(v.size)).regrid(data[0]).tolist()[:2] == [2.0, 2.0]
    with pytest.raises(TypeError):
        xa.OverlapRegridder.from_weights(ds, tgt, method=3.5)
