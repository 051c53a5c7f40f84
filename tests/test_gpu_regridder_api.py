"""
GPU (-m gpu): the Regridder classes -- the reference's user API (xugrid/regrid/regridder.py) -- on
the HIP engine.  Known answers are the numbers of the reference's own tests/fixtures
(tests/fixtures/fixture_regridder.py, tests/test_regrid/test_regridder.py), transcribed.
"""
import numpy as np
import pytest

import xugrid_amd as xa
from conftest import same_or_nan
from stepwise import host_barycentric_stepwise, host_locate_centroids_stepwise, oracle_barycentric_triplets
from xugrid_amd import meshgen

pytestmark = pytest.mark.gpu


def raster_a():  # grid_data_a: 3x3, 50 m cells, y descending
    return xa.Raster(x=[50.0, 100.0, 150.0], y=[150.0, 100.0, 50.0], dx=50.0, dy=-50.0)


def raster_b():  # grid_data_b: 4x4
    return xa.Raster(x=[25.0, 75.0, 125.0, 175.0], y=[175.0, 125.0, 75.0, 25.0], dx=50.0, dy=-50.0)


def disk_like(n=400, seed=3):
    xy, faces = meshgen.triangle_mesh(n, seed)
    return xa.Ugrid2d(xy[:, 0], xy[:, 1], -1, faces)


EXPECTED_OVERLAP = np.array([0.0, 0.5, 1.5, 2.0, 1.5, 2.0, 3.0, 3.5, 4.5, 5.0, 6.0, 6.5, 6.0, 6.5, 7.5, 8.0]).reshape(4, 4)
EXPECTED_CENTROID = np.full((4, 4), np.nan)
EXPECTED_CENTROID[1:3, 1:3] = [[0, 1], [3, 4]]
EXPECTED_LINEAR = np.full((4, 4), np.nan)
EXPECTED_LINEAR[1:3, 1:3] = [[2.0, 3.0], [5.0, 6.0]]


def test_overlap_regridder_structured_known_answer(hip):
    """expected_results_overlap (fixture_regridder.py:294-322): 3x3 -> 4x4 mean."""
    data = np.arange(9.0).reshape(3, 3)
    rg = xa.OverlapRegridder(raster_a(), raster_b(), method="mean")
    out = rg.regrid(data)
    assert out.shape == (4, 4) and out.dtype == np.float64
    np.testing.assert_allclose(out, EXPECTED_OVERLAP, rtol=1e-14)
    # layered: shape (2, 3, 3) -> (2, 4, 4); each layer equals the 2-D result (test_regridder.py:230-239)
    layered = np.arange(18.0).reshape(2, 3, 3)
    out2 = rg.regrid(layered)
    assert out2.shape == (2, 4, 4)
    assert np.array_equal(out2[0], out)
    np.testing.assert_allclose(out2[1], EXPECTED_OVERLAP + 9.0, rtol=1e-14)
    # 36 pairs of 625 m2 (tests/test_regrid/test_structured.py:108-204)
    df = rg.weights_as_dataframe()
    assert list(df.columns) == ["target_index", "source_index", "weight"]
    assert len(df) == 36 and (df["weight"] == 625.0).all()


def test_centroid_locator_known_answer(hip):
    """expected_results_centroid (fixture_regridder.py:240-291): a structured pair goes through the separable
    search, where the outer ring's centroids -- exactly ON the source boundary -- are outside (NaN) and the
    inner 2x2 targets, whose centroid is a corner shared by four source cells, get cells 0, 1, 3, 4."""
    data = np.arange(9.0).reshape(3, 3)
    out = xa.CentroidLocatorRegridder(raster_a(), raster_b()).regrid(data)
    assert same_or_nan(out, EXPECTED_CENTROID).all()
    # the same targets as quads through the polygon path (source promoted too, regridder.py:83-96): a boundary
    # point is within tolerance of an edge, so the touching source cell is used; ties -> lowest index
    quads = xa.regrid.StructuredGrid2d(raster_b()).convert_to(xa.regrid.UnstructuredGrid2d).ugrid_topology
    out = xa.CentroidLocatorRegridder(raster_a(), quads).regrid(data).reshape(4, 4)
    assert np.array_equal(out[1:3, 1:3], EXPECTED_CENTROID[1:3, 1:3])
    ring = np.ones((4, 4), dtype=bool)
    ring[1:3, 1:3] = False
    touching = np.array([[0, 0, 1, 2], [0, 0, 1, 2], [3, 3, 4, 5], [6, 6, 7, 8]], dtype=float)
    assert (np.isnan(out[ring]) | (out[ring] == touching[ring])).all()
    # strictly interior / exterior points behave as expected
    shifted = xa.Raster(x=[30.0, 80.0, 130.0, 180.0], y=[170.0, 120.0, 70.0, 20.0], dx=50.0, dy=-50.0)
    expected = np.array([[0, 1, 2, np.nan], [3, 4, 5, np.nan], [6, 7, 8, np.nan], [np.nan] * 4])
    out = xa.CentroidLocatorRegridder(raster_a(), shifted).regrid(data)
    assert same_or_nan(out, expected).all()
    quads = xa.regrid.StructuredGrid2d(shifted).convert_to(xa.regrid.UnstructuredGrid2d).ugrid_topology
    out = xa.CentroidLocatorRegridder(raster_a(), quads).regrid(data).reshape(4, 4)
    assert same_or_nan(out, expected).all()


def test_barycentric_structured_known_answer(hip):
    """expected_results_linear (fixture_regridder.py:325-362); the reference asserts its structured
    linear weights equal the unstructured barycentric ones (test_regridder.py:371-405)."""
    data = np.arange(9.0).reshape(3, 3)
    out = xa.BarycentricInterpolator(raster_a(), raster_b()).regrid(data)
    assert same_or_nan(out, EXPECTED_LINEAR).all()
    # structured == unstructured on the interior (the ring is ON the boundary: NaN in the separable path,
    # edge-interpolated values through the polygon path)
    quads = xa.regrid.StructuredGrid2d(raster_b()).convert_to(xa.regrid.UnstructuredGrid2d).ugrid_topology
    out = xa.BarycentricInterpolator(raster_a(), quads).regrid(data).reshape(4, 4)
    np.testing.assert_allclose(out[1:3, 1:3], EXPECTED_LINEAR[1:3, 1:3], rtol=1e-12)
    ring = np.ones((4, 4), dtype=bool)
    ring[1:3, 1:3] = False
    assert (np.isnan(out[ring]) | ((out[ring] >= 0.0) & (out[ring] <= 8.0))).all()
    # a target strictly inside / outside: NaN exactly where the centroid is outside the source
    shifted = xa.Raster(x=[60.0, 110.0, 160.0, 210.0], y=[140.0, 90.0, 40.0, -10.0], dx=50.0, dy=-50.0)
    bilinear = 0.8 * 0.8 * 0 + 0.2 * 0.8 * 1 + 0.8 * 0.2 * 3 + 0.2 * 0.2 * 4
    out = xa.BarycentricInterpolator(raster_a(), shifted).regrid(data)
    assert np.isnan(out[:, 3]).all() and np.isnan(out[3, :]).all() and not np.isnan(out[:3, :3]).any()
    # bilinear inside the centroid lattice: (60,140) sits 0.2 / 0.2 into the cell spanned by centroids 0,1,3,4
    np.testing.assert_allclose(out[0, 0], bilinear, rtol=1e-12)
    quads = xa.regrid.StructuredGrid2d(shifted).convert_to(xa.regrid.UnstructuredGrid2d).ugrid_topology
    out_u = xa.BarycentricInterpolator(raster_a(), quads).regrid(data).reshape(4, 4)
    assert np.isnan(out_u[:, 3]).all() and np.isnan(out_u[3, :]).all()
    np.testing.assert_allclose(out_u[0, 0], bilinear, rtol=1e-12)
    # inside the lattice of source midpoints both paths are the same bilinear interpolation
    np.testing.assert_allclose(out_u[:2, :2], out[:2, :2], rtol=1e-12)


def test_unstructured_identities(hip):
    """tests/test_regrid/test_unstructured.py:32-59 on a seeded triangle mesh."""
    grid = disk_like()
    ug = xa.regrid.UnstructuredGrid2d(grid)
    n = grid.n_face
    for relative in (True, False):
        source, target, weights = ug.overlap(ug, relative=relative)
        valid = weights > 1.0e-5 * weights.max()
        assert np.array_equal(source[valid], np.arange(n)) and np.array_equal(target[valid], np.arange(n))
        np.testing.assert_allclose(weights[valid], np.ones(n) if relative else grid.area, rtol=1e-12)
    source, target, weights = ug.locate_centroids(ug)
    assert np.array_equal(source, np.arange(n)) and np.array_equal(target, np.arange(n)) and (weights == 1).all()
    source, target, weights = ug.barycentric(ug)
    keep = weights > 1e-9
    assert np.array_equal(np.unique(target[keep]), np.arange(n))
    big = weights > 0.999999
    assert np.array_equal(source[big], target[big]) and big.sum() == n


def test_regridders_unstructured_to_structured_and_back(hip):
    """shapes and bounds as tests/test_regrid/test_regridder.py:137-201."""
    grid = disk_like(900, 5)
    z = meshgen.smooth_field(grid.centroids, 1)
    x = np.arange(0.05, 1.0, 0.1)
    target = xa.Raster(x=x, y=x[::-1].copy())
    for cls in (xa.OverlapRegridder, xa.CentroidLocatorRegridder, xa.BarycentricInterpolator, xa.RelativeOverlapRegridder):
        rg = cls(grid, target)
        out = rg.regrid(z)
        assert out.shape == (10, 10)
        layered = np.stack([z, 2 * z, 3 * z, 4 * z, 5 * z])
        out5 = rg.regrid(layered)
        assert out5.shape == (5, 10, 10) and np.array_equal(out5[0], out, equal_nan=True)
        if cls is not xa.RelativeOverlapRegridder:
            assert np.nanmin(out) >= z.min() - 1e-12 and np.nanmax(out) <= z.max() + 1e-12
        back = cls(target, grid).regrid(out)
        assert back.shape == (grid.n_face,)


def test_from_weights_and_from_dataset_round_trip(hip):
    """round trips must be exactly equal (test_regridder.py:212-257)."""
    grid = disk_like(600, 8)
    target = disk_like(500, 9)
    z = np.stack([meshgen.smooth_field(grid.centroids, k, 0.02) for k in range(3)])
    for cls, kw in ((xa.OverlapRegridder, {"method": "median"}), (xa.RelativeOverlapRegridder, {}),
                    (xa.BarycentricInterpolator, {}), (xa.CentroidLocatorRegridder, {})):
        rg = cls(grid, target, **kw) if kw else cls(grid, target)
        expected = rg.regrid(z)
        ds = rg.to_dataset()
        assert "__regrid_data" in ds and "__regrid_n" in ds and "__regrid_nnz" in ds
        again = cls.from_weights(ds, target, **kw) if kw else cls.from_weights(ds, target)
        assert np.array_equal(again.regrid(z), expected, equal_nan=True)
        again = cls.from_dataset(ds)
        if kw:
            again._setup_regrid(kw["method"])
        assert np.array_equal(again.regrid(z), expected, equal_nan=True)
    # structured target survives the round trip too
    rg = xa.OverlapRegridder(grid, raster_b())
    ds = rg.weights
    again = xa.OverlapRegridder.from_dataset(ds)
    big = xa.Raster(x=[0.2, 0.6], y=[0.6, 0.2])
    rg2 = xa.OverlapRegridder(grid, big)
    assert np.array_equal(xa.OverlapRegridder.from_dataset(rg2.weights).regrid(z), rg2.regrid(z), equal_nan=True)
    assert again.regrid(z).shape == (3, 4, 4)


def test_methods_and_errors(hip):
    grid = disk_like(300, 2)
    target = disk_like(200, 4)
    z = np.round(4 * meshgen.smooth_field(grid.centroids, 0))
    base = xa.OverlapRegridder(grid, target)
    ds = base.to_dataset()
    for method in xa.OverlapRegridder._METHODS:
        out = xa.OverlapRegridder.from_weights(ds, target, method=method).regrid(z)
        assert out.shape == (target.n_face,)
    p50 = xa.OverlapRegridder.create_percentile_method(50.0)
    a = xa.OverlapRegridder.from_weights(ds, target, method=p50).regrid(z)
    b = xa.OverlapRegridder.from_weights(ds, target, method="median").regrid(z)
    assert np.array_equal(a, b, equal_nan=True)
    with pytest.raises(ValueError):
        xa.OverlapRegridder(grid, target, method="does_not_exist")
    with pytest.raises(ValueError):
        xa.OverlapRegridder.create_percentile_method(150.0)
    # a caller's own reduction (regridder.py:136-137, examples/overlap_regridder.py:105-169): the weights are the engine's, the
    # callable runs on the host over (values, weights, workspace) of every non-empty row
    def own_mean(values, weights, workspace):
        total = 0.0
        weight_sum = 0.0
        for value, weight in zip(values, weights):
            if ~np.isnan(value):
                total += value * weight
                weight_sum += weight
        if weight_sum == 0.0:
            return np.nan
        return total / weight_sum

    zz = np.stack([z, np.where(np.arange(z.size) % 7 == 0, np.nan, z)])
    mine = xa.OverlapRegridder(grid, target, method=own_mean).regrid(zz)
    assert np.array_equal(mine, xa.OverlapRegridder(grid, target, method="mean").regrid(zz), equal_nan=True)
    assert np.array_equal(xa.OverlapRegridder(grid, target, method=lambda v, w, ws: np.nanmax(v)).regrid(z),
                          xa.OverlapRegridder(grid, target, method="maximum").regrid(z), equal_nan=True)
    with pytest.raises(TypeError):
        xa.OverlapRegridder(grid, target, method=3.5)
    with pytest.raises(TypeError):
        xa.OverlapRegridder(1, target)
    with pytest.raises(TypeError):
        base.regrid("not an array")
    with pytest.raises(ValueError):
        base.regrid(np.zeros(grid.n_face + 1))
    with pytest.raises(TypeError):
        xa.regrid.UnstructuredGrid2d(1)
    # create_percentile_method(50) on [0..4] -> 2 (test_regridder.py:282-293)
    xy = np.array([[0.0, 0], [1, 0], [2, 0], [3, 0], [4, 0], [5, 0], [0, 1], [1, 1], [2, 1], [3, 1], [4, 1], [5, 1]])
    quads = np.array([[i, i + 1, i + 7, i + 6] for i in range(5)])
    src = xa.Ugrid2d(xy[:, 0], xy[:, 1], -1, quads)
    tgt = xa.Ugrid2d(np.array([0.0, 5, 5, 0]), np.array([0.0, 0, 1, 1]), -1, np.array([[0, 1, 2, 3]]))
    out = xa.OverlapRegridder(src, tgt, method=p50).regrid(np.arange(5.0))
    assert out.shape == (1,) and out[0] == 2.0


def test_celltree_adapter_matches_numba_celltree_call_shape(hip, oracle):
    xy, faces = meshgen.triangle_mesh(500, 1)
    txy, tf = meshgen.triangle_mesh(300, 2, 20.0, 0.8)
    tree = xa.CellTree2d(xy, faces, -1)
    i, j, a = tree.intersect_faces(vertices=txy, faces=tf, fill_value=-1)
    oi, oj, oa = oracle.CellTree2d(xy, faces).intersect_faces(txy, tf)
    assert i.dtype == np.intp and j.dtype == np.intp and a.dtype == np.float64
    assert np.array_equal(i, oi) and np.array_equal(j, oj) and np.array_equal(a, oa)
    assert (np.diff(i) >= 0).all()


@pytest.mark.parametrize("kind", ["tri", "quad"])
def test_barycentric_device_pipeline(hip, oracle, kind):
    """The device-assembled CSR (xr_barycentric_csr) == the step-by-step host path == the oracle, bit for bit;
    the target sticks out of the source so that exterior Voronoi cells, replaced weights and outside points occur."""
    if kind == "tri":
        sxy, sf = meshgen.triangle_mesh(1500, 21)
    else:
        rng = np.random.default_rng(4)
        sxy, sf = meshgen.quad_mesh(np.cumsum(rng.uniform(0.5, 1.5, 31)), np.cumsum(rng.uniform(0.5, 1.5, 27)))
    txy, tf = meshgen.triangle_mesh(2500, 22)
    lo, hi = sxy.min(axis=0), sxy.max(axis=0)
    txy = lo + (hi - lo) * (0.5 + 1.12 * ((txy - txy.min(axis=0)) / (txy.max(axis=0) - txy.min(axis=0)) - 0.5))
    src = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
    tgt = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
    us, ut = xa.regrid.UnstructuredGrid2d(src), xa.regrid.UnstructuredGrid2d(tgt)
    for tol in (None, 1e-9):
        dcsr = us.barycentric_device(ut, tol)
        data, indices, indptr = dcsr.download()
        rows = np.repeat(np.arange(dcsr.n), np.diff(indptr))
        hs, ht, hw = host_barycentric_stepwise(us, ut, tol)
        assert (dcsr.n, dcsr.m) == (tgt.n_face, src.n_face)
        assert np.array_equal(indices, hs) and np.array_equal(rows, ht) and np.array_equal(data, hw)
        os_, ot, ow = oracle_barycentric_triplets(oracle, src, tgt.centroids, tol)
        assert np.array_equal(indices, os_) and np.array_equal(rows, ot) and np.array_equal(data, ow)
        outside_rows = np.diff(indptr) == 0
        assert 0 < outside_rows.sum() < tgt.n_face
        sums = np.bincount(rows, weights=data, minlength=dcsr.n)
        # rows sum to one, except in (concave) exterior Voronoi cells where negative Wachspress weights are
        # dropped by the weights > 0 filter (unstructured.py:191) -- a thin boundary layer
        assert (np.abs(sums[~outside_rows] - 1.0) < 1e-12).mean() > 0.9
    rg = xa.BarycentricInterpolator(src, tgt)
    z = 2.0 * src.centroids[:, 0] - 3.0 * src.centroids[:, 1] + 1.0
    out = rg.regrid(z)
    assert np.array_equal(np.isnan(out), outside_rows)
    # the regridder agrees with the oracle apply on the oracle's weights
    expected = oracle.regrid_csr("mean", z[None], ow, os_, oracle.to_csr_indptr(ot, tgt.n_face), tgt.n_face)[0]
    assert same_or_nan(out, expected).all()
    # explicit points instead of a query mesh
    from xugrid_amd import engine
    vg, _, _, v2f, n2n = us._voronoi()
    c2 = engine.barycentric_csr(vg.device_mesh, src.device_mesh, v2f, n2n, points=tgt.centroids)
    d2, i2, p2 = c2.download()
    assert np.array_equal(d2, data) and np.array_equal(i2, indices) and np.array_equal(p2, indptr)


def test_device_arrays_in_and_out_equal_the_host_path(hip, oracle):
    """Round 6: the reference-shaped classes reach data that already lives in HBM.  ``Ugrid2d.from_device_arrays`` + ``regrid`` of
    a device array (here the engine's own ``DeviceArray`` through ``__cuda_array_interface__``; torch tensors: the next test)
    give bit for bit what host arrays give (regridder.py:102-113, :212-262); the weights of a regridder between two device
    grids are built by its first ``regrid`` in one engine call with the apply (xr_overlap_apply_dev); the result stays on the
    device."""
    from xugrid_amd import engine

    sxy, sf = meshgen.triangle_mesh(6000, 51)
    txy, tf = meshgen.triangle_mesh(5000, 52, 30.0, 0.8)
    src_h = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
    tgt_h = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
    rng = np.random.default_rng(1)
    data = rng.normal(size=(3, sf.shape[0]))
    data[1, ::11] = np.nan
    t = engine.DeviceArray.from_host
    src_d = xa.Ugrid2d.from_device_arrays(t(sxy), t(sf))
    tgt_d = xa.Ugrid2d.from_device_arrays(t(txy), t(tf.astype(np.int32)))  # (int32 connectivity is taken as it is)
    assert (src_d.n_node, src_d.n_face, src_d.n_max_node_per_face) == (sxy.shape[0], sf.shape[0], 3)
    for cls, kwargs in ((xa.OverlapRegridder, {"method": "mean"}), (xa.OverlapRegridder, {"method": "maximum"}),
                        (xa.OverlapRegridder, {"method": "median"}), (xa.RelativeOverlapRegridder, {}),
                        (xa.CentroidLocatorRegridder, {}), (xa.BarycentricInterpolator, {})):
        expected = cls(src_h, tgt_h, **kwargs).regrid(data)
        rg = cls(src_d, tgt_d, **kwargs)
        if cls in (xa.OverlapRegridder, xa.RelativeOverlapRegridder):
            assert rg._device_weights is None and rg._deferred is not None  # built by the first regrid
        got = rg.regrid(t(data))
        assert isinstance(got, engine.DeviceArray) and got.dtype == np.float64 and got.shape == expected.shape
        assert same_or_nan(got.download(), expected).all(), cls.__name__
        assert rg._device_weights is not None
        # again (cached weights now), one variable, float32, host data on device grids
        one = rg.regrid(t(data[0]))
        assert one.shape == (tf.shape[0],) and same_or_nan(one.download(), expected[0]).all()
        d32 = data.astype(np.float32)
        assert same_or_nan(rg.regrid(t(d32)).download(), cls(src_h, tgt_h, **kwargs).regrid(d32)).all()
        assert same_or_nan(rg.regrid(data), expected).all()
        # extra leading dims are flattened and restored (regridder.py:145-163, :193-195)
        stacked = rg.regrid(t(np.stack([data, data[::-1]])))
        assert stacked.shape == (2, 3, tf.shape[0]) and same_or_nan(stacked.download()[1], expected[::-1]).all()
    # a deferred regridder asked for its weights first builds them on its own
    rg = xa.OverlapRegridder(src_d, tgt_d)
    w = rg._ensure_host_weights()
    wh = xa.OverlapRegridder(src_h, tgt_h)._ensure_host_weights()
    assert np.array_equal(w.data, wh.data) and np.array_equal(w.indices, wh.indices) and np.array_equal(w.indptr, wh.indptr)
    # the host view of a device grid is a download
    assert np.array_equal(src_d.node_coordinates, sxy) and np.array_equal(src_d.face_node_connectivity, sf)
    # a device grid paired with a host grid or a raster (no deferral then)
    mixed = xa.OverlapRegridder(src_d, tgt_h).regrid(t(data))
    assert same_or_nan(mixed.download(), xa.OverlapRegridder(src_h, tgt_h).regrid(data)).all()
    raster = xa.Raster(x=np.linspace(0.2, 0.8, 30), y=np.linspace(0.8, 0.2, 25))
    assert same_or_nan(xa.OverlapRegridder(src_d, raster).regrid(t(data)).download(), xa.OverlapRegridder(src_h, raster).regrid(data)).all()
    # errors of the host path, on device data
    with pytest.raises(ValueError):
        rg.regrid(t(data[:, :-1].copy()))
    with pytest.raises(TypeError):
        rg.regrid(t(np.arange(sf.shape[0])))
    with pytest.raises(ValueError):
        xa.Ugrid2d.from_device_arrays(t(sxy.astype(np.float32)), t(sf))
    with pytest.raises(TypeError):
        xa.Ugrid2d.from_device_arrays(sxy, sf)


def test_torch_tensors_in_and_out(tmp_path):
    """... and with torch tensors on the GPU: tensors in, a tensor out, equal to the numpy path.  torch has to initialise its HIP
    runtime BEFORE the engine binds the device, so this runs in a process of its own (tests/device_api_worker_gpu.py)."""
    import os
    import subprocess
    import sys

    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "device_api_worker_gpu.py")
    res = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "TORCH_DEVICE_API_OK" in res.stdout


def test_deferred_points_survive_invalidate_and_release_of_their_meshes(hip):
    """xr_locate_flags_begin defers its kernels (round 5); the handle keeps raw mesh pointers.  A source mesh INVALIDATED and a
    query mesh released between the handle's creation and its use must not change the result: xr_mesh_invalidate /
    xr_mesh_destroy launch the pending kernels first, the Python handle keeps both meshes alive."""
    import gc

    from xugrid_amd import engine

    sxy, sf = meshgen.triangle_mesh(3000, 31)
    txy, tf = meshgen.triangle_mesh(5000, 32, 20.0, 0.9)
    src = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
    tgt = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
    us = xa.regrid.UnstructuredGrid2d(src)
    expected = us.barycentric_device(xa.regrid.UnstructuredGrid2d(tgt)).download()
    voronoi_mesh, tail, n2n = us._voronoi_device()
    for mode in ("invalidate_source", "drop_query", "both"):
        query_mesh = engine.DeviceMesh(txy, tf, -1)
        prepared = engine.DevicePoints(src.device_mesh, query=query_mesh)
        if mode in ("invalidate_source", "both"):
            src.device_mesh.invalidate()  # releases the index the deferred locate pass reads
        if mode in ("drop_query", "both"):
            del query_mesh
            gc.collect()
        got = engine.barycentric_csr(voronoi_mesh, src.device_mesh, tail, n2n, n_identity=src.n_face, prepared=prepared).download()
        for a, b in zip(got, expected):
            assert np.array_equal(a, b), mode
    # a handle that is never consumed is released without its kernels having run
    lonely = engine.DevicePoints(src.device_mesh, query=engine.DeviceMesh(txy, tf, -1))
    del lonely
    gc.collect()


def test_barycentric_concave_reference_known_answer(hip, oracle):
    """The reference's test_barycentric_concave (tests/test_regrid/test_regridder.py:334-369) through the public classes
    on the device: BarycentricInterpolator(Ugrid2d, Raster) -> exactly 200 NaN cells, 0.5 <= v <= 2.0; and the device
    weights == the oracle's step-by-step weights, bit for bit."""
    from stepwise import CONCAVE_FACES, CONCAVE_VALUES, CONCAVE_VERTICES, concave_raster_axes

    grid = xa.Ugrid2d(CONCAVE_VERTICES[:, 0], CONCAVE_VERTICES[:, 1], -1, CONCAVE_FACES)
    x, y = concave_raster_axes()
    regridder = xa.BarycentricInterpolator(source=grid, target=xa.Raster(x=x, y=y))
    result = regridder.regrid(CONCAVE_VALUES)
    assert result.shape == (y.size, x.size) == (20, 30)
    assert np.nanmin(result) >= 0.5 and np.nanmax(result) <= 2.0
    assert np.isnan(result).sum() == 200
    # the query points are the centroids of the raster's cells as quadrilaterals (unstructured.py:147), row-major (y, x)
    quads = xa.regrid.StructuredGrid2d(xa.Raster(x=x, y=y)).convert_to(xa.regrid.UnstructuredGrid2d).ugrid_topology
    points = oracle.centroids(quads.node_coordinates, quads.face_node_connectivity)
    yy, xx = np.meshgrid(y, x, indexing="ij")
    assert np.allclose(points, np.column_stack([xx.ravel(), yy.ravel()]), rtol=0, atol=1e-12)
    os_, ot, ow = oracle_barycentric_triplets(oracle, grid, points)
    w = regridder._ensure_host_weights()
    rows = np.repeat(np.arange(w.n), np.diff(w.indptr))
    assert np.array_equal(w.indices, os_) and np.array_equal(rows, ot) and np.array_equal(w.data, ow)
    expected = oracle.regrid_csr("mean", CONCAVE_VALUES[None, :], ow, os_, oracle.to_csr_indptr(ot, points.shape[0]), points.shape[0])[0]
    assert same_or_nan(result.ravel(), expected).all()
    # a descending-y raster (the usual DataArray orientation) is the same picture upside down (the cells' centroids come from
    # the vertices in another order: equal up to their rounding)
    flipped = xa.BarycentricInterpolator(source=grid, target=xa.Raster(x=x, y=y[::-1].copy())).regrid(CONCAVE_VALUES)
    assert np.array_equal(np.isnan(flipped), np.isnan(result[::-1])) and np.isnan(flipped).sum() == 200
    np.testing.assert_allclose(flipped, result[::-1], rtol=1e-12, atol=0, equal_nan=True)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_voronoi_device_matches_reference_goldens(hip, golden, tag):
    """G6: voronoi_topology of the reference on seeded Delaunay meshes (50 / 500 / 5000 points).  The device
    pre-step (xr_voronoi_*) + host boundary cells must give the same tessellation, array for array."""
    from xugrid_amd import engine, voronoi

    g = golden("g6_voronoi.npz")
    xy, faces = g[tag + "_xy"], g[tag + "_faces"]
    grid = xa.Ugrid2d(xy[:, 0], xy[:, 1], -1, faces)
    # the device's node -> face inversion and exterior edges == the host connectivities
    b = engine.DeviceVoronoi(grid.device_mesh)
    indptr, indices, edge_nodes, edge_face, cen = b.download()
    nfc = grid.node_face_connectivity.tocsr()
    assert np.array_equal(indptr, nfc.indptr) and np.array_equal(indices, nfc.indices)
    enc, efc = g[tag + "_enc"], g[tag + "_efc"]
    ext = efc[:, 1] == -1
    assert np.array_equal(edge_nodes, enc[ext]) and np.array_equal(edge_face, efc[ext, 0])
    assert np.array_equal(cen, grid.centroids)
    np.testing.assert_allclose(cen, g[tag + "_centroids"], rtol=1e-15)
    mesh, face_i, nmap = voronoi.voronoi_topology_device(grid)
    v, f = mesh.download()
    # vertices: the reference's centroids are numpy nanmean, ours the device kernel's (sum / 3): last-ulp apart
    np.testing.assert_allclose(v, g[tag + "_vor_vertices"], rtol=1e-14, atol=1e-16)
    gf = g[tag + "_vor_faces"]
    assert f.shape[1] <= gf.shape[1] or (f[:, gf.shape[1]:] == -1).all()
    k = min(f.shape[1], gf.shape[1])
    assert np.array_equal(f[:, :k], gf[:, :k]) and (gf[:, k:] == -1).all()
    assert np.array_equal(face_i, g[tag + "_vor_face_i"])
    assert np.array_equal(np.sort(nmap, axis=1), np.sort(g[tag + "_vor_nmap"], axis=1))
    # and it is exactly what the all-host function produces from the same (device) centroids
    hv, hf, hfi, hnm = voronoi.voronoi_topology(
        grid.node_face_connectivity, grid.node_coordinates, grid.centroids, grid.edge_face_connectivity,
        grid.edge_node_connectivity, add_exterior=True, add_vertices=True, skip_concave=True,
    )
    assert np.array_equal(v, hv) and np.array_equal(f, hf) and np.array_equal(face_i, hfi) and np.array_equal(nmap, hnm)


def test_voronoi_device_mixed_and_large(hip):
    """quads (+ a mixed tri/quad mesh) and a 200k-face lattice mesh: device tessellation == host tessellation."""
    from xugrid_amd import voronoi

    rng = np.random.default_rng(8)
    qxy, qf = meshgen.quad_mesh(np.cumsum(rng.uniform(0.5, 1.5, 40)), np.cumsum(rng.uniform(0.5, 1.5, 33)))
    # mixed: split the first 300 quads into triangles, pad with -1
    tri_a = np.column_stack([qf[:300, 0], qf[:300, 1], qf[:300, 2], np.full(300, -1)])
    tri_b = np.column_stack([qf[:300, 0], qf[:300, 2], qf[:300, 3], np.full(300, -1)])
    mixed = np.vstack([tri_a, tri_b, qf[300:]])
    lxy, lf = meshgen.triangle_mesh(100_000, 5, delaunay=False)
    dxy, df = meshgen.triangle_mesh(30_000, 6)  # Delaunay: concave exterior cells, hull slivers
    for xy, faces in ((qxy, qf), (qxy, mixed), (lxy, lf), (dxy, df)):
        grid = xa.Ugrid2d(xy[:, 0], xy[:, 1], -1, faces)
        mesh, face_i, nmap = voronoi.voronoi_topology_device(grid)  # boundary cells by the library (native code)
        v, f = mesh.download()
        # ... and with the numpy restatement of the boundary part: the same tessellation bit for bit (straight lattice
        # boundaries included, where the convexity choice hangs on the summation order of the polygon areas)
        mesh_h, face_h, nmap_h = voronoi.voronoi_topology_device(grid, host_boundary=True)
        v_h, f_h = mesh_h.download()
        assert np.array_equal(v, v_h) and np.array_equal(f, f_h)
        assert np.array_equal(face_i, face_h) and np.array_equal(nmap, nmap_h)
        # the boundary cells as the C ABI hands them out (xr_voronoi_boundary_cells): the tail of the tessellation
        from xugrid_amd import engine

        extra, bcells = engine.DeviceVoronoi(grid.device_mesh).boundary_cells()
        assert np.array_equal(extra, v[grid.n_face:])
        nb = bcells.shape[0]
        k = min(bcells.shape[1], f.shape[1])
        assert np.array_equal(bcells[:, :k], f[f.shape[0] - nb:, :k]) and (bcells[:, k:] == -1).all()
        hv, hf, hfi, hnm = voronoi.voronoi_topology(
            grid.node_face_connectivity, grid.node_coordinates, grid.centroids, grid.edge_face_connectivity,
            grid.edge_node_connectivity, add_exterior=True, add_vertices=True, skip_concave=True,
        )
        assert np.array_equal(v, hv) and np.array_equal(f, hf)
        assert np.array_equal(face_i, hfi) and np.array_equal(nmap, hnm)


def test_centroid_locator_device_pipeline(hip, oracle):
    """xr_locate_csr + the select apply == locate_points + the COO scatter of the oracle (regridder.py:386-409)."""
    sxy, sf = meshgen.triangle_mesh(2000, 31)
    txy, tf = meshgen.triangle_mesh(3000, 32)
    txy = 0.5 + 1.1 * (txy - 0.5)  # part of the target lies outside the source
    src = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
    tgt = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
    rg = xa.CentroidLocatorRegridder(src, tgt)
    tree = oracle.CellTree2d(sxy, sf, -1)
    located = tree.locate_points(oracle.centroids(txy, tf))
    row = np.nonzero(located >= 0)[0]
    col = located[row]
    assert 0 < row.size < tgt.n_face
    df = rg.weights_as_dataframe()
    assert np.array_equal(df["target_index"], row) and np.array_equal(df["source_index"], col)
    assert (df["weight"] == 1.0).all()
    rng = np.random.default_rng(0)
    data = rng.normal(size=(5, src.n_face))
    data[1, ::7] = np.nan  # NaN is copied, not skipped
    data[2, ::5] = -0.0    # and so is the sign of zero
    data32 = data.astype(np.float32)
    for d in (data, data32):
        out = rg.regrid(d)
        expected = oracle.regrid_coo(d, row, col, tgt.n_face)
        assert out.dtype == np.float64 and same_or_nan(out, expected).all()
        assert np.array_equal(np.signbit(out), np.signbit(expected))
    # many variables take the planned apply kernels
    many = rng.normal(size=(40, src.n_face))
    assert same_or_nan(rg.regrid(many), oracle.regrid_coo(many, row, col, tgt.n_face)).all()
    # weights round trip (MatrixCOO fields) and unsorted external COO weights
    rg2 = xa.CentroidLocatorRegridder.from_dataset(rg.to_dataset())
    assert same_or_nan(rg2.regrid(data), rg.regrid(data)).all()
    ds = rg.to_dataset()
    perm = rng.permutation(row.size)
    for key in ("__regrid_data", "__regrid_row", "__regrid_col"):
        ds[key] = np.asarray(ds[key])[perm]
    rg3 = xa.CentroidLocatorRegridder.from_weights(ds, tgt)
    assert same_or_nan(rg3.regrid(data), rg.regrid(data)).all()
    # explicit points
    from xugrid_amd import engine
    c = engine.locate_csr(src.device_mesh, points=tgt.centroids)
    d1, i1, p1 = c.download()
    d0, i0, p0 = rg._device_weights.download()
    assert np.array_equal(i1, i0) and np.array_equal(p1, p0) and np.array_equal(d1, d0)


def test_rasterize_known_answers(hip):
    """tests/test_ugrid2d.py:794-825 (numbers transcribed): two quads under two triangles."""
    vertices = np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0], [0.0, 1.0], [1.0, 1.0], [2.0, 1.0], [1.0, 2.0]])
    faces = np.array([[0, 1, 4, 3], [1, 2, 5, 4], [3, 4, 6, -1], [4, 5, 6, -1]])
    grid = xa.Ugrid2d(vertices[:, 0], vertices[:, 1], -1, faces)
    x, y, index = grid.rasterize(resolution=0.5)
    assert np.allclose(x, [0.25, 0.75, 1.25, 1.75]) and np.allclose(y, [1.75, 1.25, 0.75, 0.25])
    assert np.array_equal(index, [[-1, 2, 3, -1], [2, 2, 3, 3], [0, 0, 1, 1], [0, 0, 1, 1]])
    x, y, index = grid.rasterize(resolution=0.5, bounds=(-1.0, -1.0, 2.0, 2.0))
    expected = np.array(
        [
            [-1, -1, -1, 2, 3, -1],
            [-1, -1, 2, 2, 3, 3],
            [-1, -1, 0, 0, 1, 1],
            [-1, -1, 0, 0, 1, 1],
            [-1, -1, -1, -1, -1, -1],
            [-1, -1, -1, -1, -1, -1],
        ]
    )
    assert np.allclose(x, [-0.75, -0.25, 0.25, 0.75, 1.25, 1.75]) and np.allclose(y, [1.75, 1.25, 0.75, 0.25, -0.25, -0.75])
    assert np.array_equal(index, expected)
    xs, ys, idx = grid.rasterize_like(np.array([0.5, 1.5]), np.array([0.5]))
    assert np.array_equal(idx, [[0, 1]])
    # the raster's nodes are generated on the device: same answer as locate_points on the host-built meshgrid
    sxy, sf = meshgen.triangle_mesh(4000, 5)
    big = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
    rx, ry = np.linspace(-0.1, 1.1, 173), np.linspace(1.05, -0.05, 131)
    _, _, idx = big.rasterize_like(rx, ry)
    yy, xx = np.meshgrid(ry, rx, indexing="ij")
    assert np.array_equal(idx, big.locate_points(np.column_stack([xx.ravel(), yy.ravel()])).reshape(131, 173))
    assert big.rasterize_like(np.zeros(0), ry)[2].shape == (131, 0)


def test_barycentric_full_size_vs_oracle(hip, oracle):
    """BASELINE config 3 at 1M source faces / 1M query points: device Voronoi pre-step + xr_barycentric_csr against
    the oracle's step-by-step restatement -- identical triplets."""
    sxy, sf = meshgen.triangle_mesh(500_000, 0, delaunay=False)
    txy, tf = meshgen.triangle_mesh(500_000, 2, 30.0, 0.75, delaunay=False)
    src = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
    tgt = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
    dcsr = xa.regrid.UnstructuredGrid2d(src).barycentric_device(xa.regrid.UnstructuredGrid2d(tgt))
    data, indices, indptr = dcsr.download()
    rows = np.repeat(np.arange(dcsr.n), np.diff(indptr))
    os_, ot, ow = oracle_barycentric_triplets(oracle, src, oracle.centroids(txy, tf))
    assert indices.size == os_.size > 4_000_000
    assert np.array_equal(indices, os_) and np.array_equal(rows, ot) and np.array_equal(data, ow)


def test_weights_file_round_trip(hip, tmp_path):
    """to_file / from_file (npz of the reference's dataset variables) for unstructured and structured regridders."""
    grid, target = disk_like(700, 3), disk_like(650, 4)
    z = np.stack([meshgen.smooth_field(grid.centroids, k, 0.02) for k in range(2)])
    cases = [
        (xa.OverlapRegridder(grid, target, method="mean"), z),
        (xa.BarycentricInterpolator(grid, target), z),
        (xa.CentroidLocatorRegridder(grid, target), z),
        (xa.RelativeOverlapRegridder(grid, raster_b()), z),
        (xa.OverlapRegridder(raster_a(), raster_b()), np.arange(18.0).reshape(2, 3, 3)),
    ]
    for i, (rg, data) in enumerate(cases):
        path = tmp_path / f"weights_{i}.npz"
        rg.to_file(path)
        again = type(rg).from_file(path)
        assert np.array_equal(again.regrid(data), rg.regrid(data), equal_nan=True)


def test_rectilinear_mesh_generated_on_device(hip):
    """xr_mesh_create_rectilinear == Ugrid2d.from_structured_bounds (ugrid2d.py:1973-2034) for every combination
    of ascending / descending axes; the regridders take rasters through it without host copies of the quads."""
    from xugrid_amd.engine import DeviceMesh
    from xugrid_amd.regrid.structured import StructuredGrid2d
    from xugrid_amd.ugrid2d import RectilinearUgrid2d

    rng = np.random.default_rng(5)
    for flip_x in (False, True):
        for flip_y in (False, True):
            xv = np.concatenate(([0.0], np.cumsum(rng.uniform(0.5, 2.0, 37))))
            yv = np.concatenate(([-3.0], -3.0 + np.cumsum(rng.uniform(0.5, 2.0, 23))))
            xv, yv = (xv[::-1].copy() if flip_x else xv), (yv[::-1].copy() if flip_y else yv)
            x_bounds = np.column_stack([np.minimum(xv[:-1], xv[1:]), np.maximum(xv[:-1], xv[1:])])
            y_bounds = np.column_stack([np.minimum(yv[:-1], yv[1:]), np.maximum(yv[:-1], yv[1:])])
            host = xa.Ugrid2d.from_structured_bounds(x_bounds, y_bounds)
            dev = xa.Ugrid2d.from_structured_bounds_device(x_bounds, y_bounds)
            assert isinstance(dev, RectilinearUgrid2d) and dev._host is None
            assert (dev.n_node, dev.n_face, dev.n_max_node_per_face) == (host.n_node, host.n_face, 4)
            xy, faces = dev.device_mesh.download()
            assert dev._host is None  # nothing was materialised on the host
            assert np.array_equal(xy, host.node_coordinates) and np.array_equal(faces, host.face_node_connectivity)
            assert np.array_equal(dev.area, host.area) and np.array_equal(dev.centroids, host.centroids)
            assert np.array_equal(dev.face_node_connectivity, host.face_node_connectivity)  # lazy host copy
    # a raster target through the regridder: same weights as with the host-built quads
    sxy, sf = meshgen.triangle_mesh(3000, 0)
    src = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
    raster = xa.Raster(x=np.linspace(0.05, 0.95, 40), y=np.linspace(0.9, 0.1, 33))
    rg = xa.OverlapRegridder(src, raster, method="mean")
    quads = xa.Ugrid2d.from_structured_bounds(
        StructuredGrid2d(raster).xbounds.directional_bounds, StructuredGrid2d(raster).ybounds.directional_bounds
    )
    ref = xa.OverlapRegridder(src, quads, method="mean")
    a, b = rg.weights_as_dataframe(), ref.weights_as_dataframe()
    assert a.equals(b)
    data = np.random.default_rng(0).normal(size=src.n_face)
    assert np.array_equal(rg.regrid(data).ravel(), ref.regrid(data), equal_nan=True)
    with pytest.raises(ValueError):
        DeviceMesh.from_rectilinear(np.array([0.0, np.nan, 2.0]), np.array([0.0, 1.0]))
    with pytest.raises(ValueError):
        DeviceMesh.from_rectilinear(np.array([0.0]), np.array([0.0, 1.0]))


def test_barycentric_from_raster_source_paired_with_mesh(hip):
    """A raster source paired with an unstructured target is promoted to device-generated quads; the Voronoi
    pre-step then only needs the coordinates of the boundary nodes, which come from the two 1-D vertex arrays --
    the host copies of the quads are never made.  Same weights as with host-built quads."""
    from xugrid_amd.regrid.structured import StructuredGrid2d

    raster = xa.Raster(x=np.linspace(0.0, 1.0, 30), y=np.linspace(1.0, 0.0, 25))
    txy, tf = meshgen.triangle_mesh(2000, 4, 10.0, 0.8)
    tgt = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
    rg = xa.BarycentricInterpolator(raster, tgt)
    assert rg._source._unstructured.ugrid_topology._host is None
    grid = StructuredGrid2d(raster)
    quads = xa.Ugrid2d.from_structured_bounds(grid.xbounds.directional_bounds, grid.ybounds.directional_bounds)
    ref = xa.BarycentricInterpolator(quads, tgt)
    assert rg.weights_as_dataframe().equals(ref.weights_as_dataframe())


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_replace_interpolated_weights_device_golden(hip, golden, tag):
    """G10 (reference output of unstructured.py:17-57) through the device kernel of the barycentric pipeline."""
    g = golden("g10_replace.npz")
    w = g[f"{tag}_weights_in"].copy()
    hip.engine.replace_interpolated_weights(g[f"{tag}_vertices"], g[f"{tag}_faces"], g[f"{tag}_face_index"], w,
                                            g[f"{tag}_node_to_node_map"], int(g[f"{tag}_threshold"]))
    assert np.array_equal(w, g[f"{tag}_weights_out"])


def _reversed_cells(voronoi_mesh):
    """Voronoi cells the tree stores in another vertex order than the caller's (download = caller's order)."""
    _, faces = voronoi_mesh.download()
    ccw = voronoi_mesh.faces_ccw()
    k = min(faces.shape[1], ccw.shape[1])
    return int((faces[:, :k] != ccw[:, :k]).any(axis=1).sum())


@pytest.mark.parametrize("case", ["g6c", "delaunay_100k", "clockwise_source"])
def test_barycentric_tree_order_blast_radius(hip, oracle, golden, case):
    """the weight slots of a Voronoi cell are paired with the caller's vertex order by default (the
    reference's pairing, unstructured.py:175,193) and with the tree's counter-clockwise vertex order under the opt-in
    tree_order=True.  The two can only differ for cells the tree stores reversed.  MEASURED here (printed; the
    numbers are quoted in DESIGN.md section 7): how many cells that is -- concave exterior cells only, a fraction of
    the boundary -- and how many (source, target) entries change.  The host step-by-step path equals the device
    pipeline entry for entry under BOTH settings; entries only differ when reversed cells exist."""
    from xugrid_amd import voronoi

    if case == "g6c":
        g = golden("g6_voronoi.npz")
        xy, faces = g["c_xy"], g["c_faces"]
    else:
        xy, faces = meshgen.triangle_mesh(50_000, 5)
        if case == "clockwise_source":
            faces = faces[:, ::-1].copy()
    rng = np.random.default_rng(8)
    lo, hi = xy.min(axis=0), xy.max(axis=0)
    points = lo + (hi - lo) * rng.random((40_000, 2))
    src = xa.Ugrid2d(xy[:, 0], xy[:, 1], -1, faces)
    tgt_xy, tgt_f = meshgen.triangle_mesh(20_000, 6, 0.0, 1.0, delaunay=False)
    tgt_xy = lo + (hi - lo) * tgt_xy
    tgt = xa.Ugrid2d(tgt_xy[:, 0], tgt_xy[:, 1], -1, tgt_f)
    mesh, _, _ = voronoi.voronoi_topology_device(src)
    n_rev = _reversed_cells(mesh)
    a = xa.regrid.UnstructuredGrid2d(src)
    b = xa.regrid.UnstructuredGrid2d(tgt)
    trip = {}
    for ref in (False, True):
        s_i, t_i, w = host_barycentric_stepwise(a, b, tree_order=ref)
        d = a.barycentric_device(b, tree_order=ref)
        data, idx, indptr = d.download()
        rows = np.repeat(np.arange(d.n), np.diff(indptr))
        assert np.array_equal(idx, s_i) and np.array_equal(rows, t_i) and np.array_equal(data, w), (case, ref)
        o_s, o_t, o_w = oracle_barycentric_triplets(oracle, src, tgt.centroids, tree_order=ref)
        assert np.array_equal(idx, o_s) and np.array_equal(rows, o_t) and np.array_equal(data, o_w), (case, ref)
        trip[ref] = (s_i, t_i, w)
    same = all(np.array_equal(x, y) for x, y in zip(trip[False], trip[True]))
    n_diff = 0 if same else int(np.setxor1d(trip[False][0] * (tgt_f.shape[0] + 1) + trip[False][1],
                                            trip[True][0] * (tgt_f.shape[0] + 1) + trip[True][1]).size)
    print(f"[blast radius] {case}: {n_rev} reversed Voronoi cells of {mesh.n_face}, {n_diff} (source, target) entries differ "
          f"of {trip[False][0].size}")
    assert (n_diff == 0) == same and (n_rev > 0 or same), (case, n_rev, n_diff)
    assert n_rev <= 0.02 * mesh.n_face  # exterior cells only
    assert n_diff <= 0.05 * trip[False][0].size
    del points


def test_concurrent_applies_from_threads(hip, oracle):
    """dask's threaded scheduler calls ``_regrid`` from several threads at once (regridder.py:177-185): the apply
    entry points run under the engine's SHARED scope, each thread on a lane of its own.  Four threads hammer four
    different weight matrices (and one shared one) with different reducers and K; every result must equal the
    single-threaded one bit for bit."""
    import threading

    from xugrid_amd import engine as E

    rng = np.random.default_rng(12)
    cases = []
    for i in range(4):
        sxy, sf = meshgen.triangle_mesh(6000 + 1500 * i, 20 + i)
        txy, tf = meshgen.triangle_mesh(5000 + 1000 * i, 30 + i, 30.0, 0.7)
        csr = E.DeviceMesh(sxy, sf).overlap(E.DeviceMesh(txy, tf))
        K = (1, 3, 12, 40)[i]
        v = rng.normal(size=(K, csr.m))
        v[0, ::9] = np.nan
        mid = (0, 5, 0, 3)[i]
        cases.append((csr, v, mid, csr.apply(v, mid)))
    shared_csr, shared_v, _, _ = cases[2]
    shared_exp = shared_csr.apply(shared_v, 9)
    errors = []

    def worker(i):
        try:
            csr, v, mid, exp = cases[i]
            for it in range(25):
                got = csr.apply(v, mid)
                if not same_or_nan(got, exp).all():
                    errors.append((i, it, "own"))
                    return
                if it % 5 == 0 and not same_or_nan(shared_csr.apply(shared_v, 9), shared_exp).all():
                    errors.append((i, it, "shared"))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # building weights (exclusive scope) while other threads apply
    def builder():
        try:
            sxy, sf = meshgen.triangle_mesh(4000, 50)
            txy, tf = meshgen.triangle_mesh(3000, 51, 30.0, 0.7)
            for _ in range(5):
                c = E.DeviceMesh(sxy, sf).overlap(E.DeviceMesh(txy, tf))
                assert c.nnz > 0
        except Exception as e:  # noqa: BLE001
            errors.append(("builder", repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(3)] + [threading.Thread(target=builder)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_concurrent_applies_in_async_mode(hip, oracle):
    """Asynchronous mode (``xr_set_async``) keeps every call on the ONE main stream and hands out no lanes; the stream's fork / join
    state is not atomic, so shared-scope calls from several host threads take turns (``SharedScope``, ``xr_engine.hip``).  Four
    threads apply matrices WITH long rows (fine -> coarse: the many-variable path forks its long rows to the side stream) at K >= 8:
    every result equals the single-threaded one bit for bit."""
    import threading

    from xugrid_amd import engine as E

    rng = np.random.default_rng(21)
    cases = []
    for i in range(4):
        sxy, sf = meshgen.triangle_mesh(30000 + 4000 * i, 60 + i)
        txy, tf = meshgen.triangle_mesh(300 + 40 * i, 70 + i, 30.0, 0.7)  # ~100 source faces per target: rows beyond 32 entries
        csr = E.DeviceMesh(sxy, sf).overlap(E.DeviceMesh(txy, tf))
        K = (8, 12, 16, 40)[i]
        v = rng.normal(size=(K, csr.m))
        v[0, ::11] = np.nan
        cases.append((csr, v, 0, csr.apply(v, 0)))
    errors = []

    def worker(i):
        try:
            csr, v, mid, exp = cases[i]
            for it in range(20):
                if not same_or_nan(csr.apply(v, mid), exp).all():
                    errors.append((i, it))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    E.set_async(True)
    try:
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        E.set_async(False)
    assert not errors, errors


def test_config1_as_survey_words_it(hip, oracle, golden):
    """BASELINE config 1 exactly as SURVEY 8(d) states it, through the public API: the committed elevation_nl arrays as a ``Ugrid2d``
    -> a 200 x 200 raster with x ASCENDING and y DESCENDING (``dy`` negative, the ``fixture_regridder.py:94-104`` style),
    ``OverlapRegridder(..., "mean").regrid(elevation float32)`` -> (200, 200) float64, NaN where nothing overlaps; sum of all
    weights = the mesh area 4.2169478944e10 m2 (1e-10 rel), every row sum <= the cell area, values within [-60.66, 252.73], and
    equal to the oracle on the same quads."""
    g = golden("g8_elevation_nl.npz")
    node_x, node_y, faces, elev = g["node_x"], g["node_y"], g["face_nodes"].astype(np.int64), g["elevation"]
    assert elev.dtype == np.float32 and faces.shape == (5248, 3) and node_x.size == 2790
    xmin, xmax, ymin, ymax = node_x.min(), node_x.max(), node_y.min(), node_y.max()
    n = 200
    dx, dy = (xmax - xmin) / n, (ymax - ymin) / n
    x = xmin + (np.arange(n) + 0.5) * dx
    y = ymax - (np.arange(n) + 0.5) * dy
    assert (np.diff(x) > 0).all() and (np.diff(y) < 0).all()
    source = xa.Ugrid2d(node_x, node_y, -1, faces)
    target = xa.Raster(x=x, y=y, dx=dx, dy=-dy)
    rg = xa.OverlapRegridder(source, target, method="mean")
    out = rg.regrid(elev)
    assert out.shape == (n, n) and out.dtype == np.float64
    w = rg._ensure_host_weights()
    assert w.n == n * n and w.m == 5248
    # the documented checks
    assert abs(w.data.sum() / 4.2169478944e10 - 1.0) < 1e-10
    assert abs(w.data.sum() / source.area.sum() - 1.0) < 1e-10
    row_sum = np.bincount(np.repeat(np.arange(w.n), np.diff(w.indptr)), weights=w.data, minlength=w.n)
    assert (row_sum <= dx * dy * (1 + 1e-12)).all()
    empty = np.diff(w.indptr) == 0
    assert np.array_equal(np.isnan(out).ravel(), empty) and empty.any() and not empty.all()
    assert np.nanmin(out) >= -60.67 and np.nanmax(out) <= 252.74
    # the raster's cells as the quads the regridder clipped: row-major (y, x) numbering, y descending
    quads = xa.regrid.StructuredGrid2d(target).convert_to(xa.regrid.UnstructuredGrid2d).ugrid_topology
    qxy, qf = np.asarray(quads.node_coordinates), np.asarray(quads.face_node_connectivity)
    cen = qxy[qf].mean(axis=1).reshape(n, n, 2)
    np.testing.assert_allclose(cen[..., 0], np.broadcast_to(x[None, :], (n, n)), rtol=1e-12)
    np.testing.assert_allclose(cen[..., 1], np.broadcast_to(y[:, None], (n, n)), rtol=1e-12)
    # ... and the oracle on those quads
    sxy = np.column_stack([node_x, node_y])
    oq, os_, oa = oracle.CellTree2d(sxy, faces).intersect_faces(qxy, qf)
    assert np.array_equal(np.repeat(np.arange(w.n), np.diff(w.indptr)), oq) and np.array_equal(w.indices, os_)
    assert np.array_equal(w.data, oa)
    exp = oracle.regrid_csr("mean", elev[None, :].astype(np.float64), w.data, w.indices, w.indptr, w.n)[0]
    long_rows = np.diff(w.indptr) > 32
    assert same_or_nan(out.ravel()[~long_rows], exp[~long_rows]).all()
    np.testing.assert_allclose(out.ravel(), exp, rtol=1e-13, equal_nan=True)
    # the value at a raster cell is the area-weighted mean of the triangles under it: spot-check the cell with the most entries
    t = int(np.argmax(np.diff(w.indptr)))
    sl = slice(w.indptr[t], w.indptr[t + 1])
    np.testing.assert_allclose(out.ravel()[t], (w.data[sl] * elev[w.indices[sl]].astype(np.float64)).sum() / w.data[sl].sum(), rtol=1e-12)


def test_barycentric_rows_longer_than_the_guess(hip, oracle):
    """The barycentric CSR is filled into arrays sized by a guess (seven entries per point) before the host knows nnz; a matrix
    that does not fit is filled again into arrays of its real size.  A fan of 16 triangles around one node: the node's Voronoi
    cell has 16 corners, and 30 000 points inside it carry 16 weights each -- more than twice the guess.  Device == oracle."""
    from xugrid_amd import engine

    n_sector = 16
    ang = 2 * np.pi * np.arange(n_sector) / n_sector
    ring1 = np.column_stack([np.cos(ang), np.sin(ang)])
    ring2 = 2.0 * np.column_stack([np.cos(ang + 0.1), np.sin(ang + 0.1)])
    xy = np.vstack([[0.0, 0.0], ring1, ring2])
    faces = []
    for i in range(n_sector):
        j = (i + 1) % n_sector
        faces.append([0, 1 + i, 1 + j])
        faces.append([1 + i, 1 + n_sector + i, 1 + n_sector + j])
        faces.append([1 + i, 1 + n_sector + j, 1 + j])
    faces = np.array(faces, dtype=np.int64)
    src = xa.Ugrid2d(xy[:, 0], xy[:, 1], -1, faces)
    rng = np.random.default_rng(3)
    r, a = 0.3 * np.sqrt(rng.random(30_000)), rng.uniform(0, 2 * np.pi, 30_000)
    pts = np.ascontiguousarray(np.column_stack([r * np.cos(a), r * np.sin(a)]))
    us = xa.regrid.UnstructuredGrid2d(src)
    voronoi_mesh, face_index_tail, n2n = us._voronoi_device()
    c = engine.barycentric_csr(voronoi_mesh, src.device_mesh, face_index_tail, n2n, points=pts, n_identity=src.n_face)
    data, indices, indptr = c.download()
    assert c.nnz > 7 * pts.shape[0] + (1 << 16), "the case no longer exceeds the guess"
    os_, ot, ow = oracle_barycentric_triplets(oracle, src, pts)
    rows = np.repeat(np.arange(c.n), np.diff(indptr))
    assert np.array_equal(indices, os_) and np.array_equal(rows, ot) and np.array_equal(data, ow)
