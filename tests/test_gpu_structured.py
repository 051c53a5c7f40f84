"""
GPU (-m gpu): the separable structured -> structured path.  The device assembles the CSR
of the per-axis outer product (xr_csr_from_outer); it must equal, bit for bit, the reference's
broadcast + argsort triplets (golden vectors) and the host restatement, and the regridders built on it must
agree with the oracle apply and with the polygon-clip path on the same rasters.
"""
import numpy as np
import pytest

import xugrid_amd as xa
from conftest import same_or_nan
from structured_cases import canon, golden_triplets, oracle_axes, random_raster, raster_kwargs
from xugrid_amd.regrid.structured import Raster, StructuredGrid2d

pytestmark = pytest.mark.gpu

CASES = list("abcdefghi")


def csr_triplets(dcsr):
    data, indices, indptr = dcsr.download()
    rows = np.repeat(np.arange(dcsr.n), np.diff(indptr))
    return indices, rows, data, indptr


@pytest.mark.parametrize("name", CASES)
def test_outer_csr_matches_reference_goldens(hip, golden, name):
    g = golden("g9_structured.npz")
    s = StructuredGrid2d(Raster(**raster_kwargs(g, name, "src")))
    t = StructuredGrid2d(Raster(**raster_kwargs(g, name, "tgt")))
    for kind, dcsr in (
        ("overlap", s.overlap_device(t, False)),
        ("relative", s.overlap_device(t, True)),
        ("linear", s.linear_weights_device(t)),
    ):
        gs, gt, gw = golden_triplets(g, name, kind)
        assert (dcsr.n, dcsr.m, dcsr.nnz) == (t.size, s.size, gs.size)
        cols, rows, data, indptr = csr_triplets(dcsr)
        assert np.array_equal(indptr, np.concatenate(([0], np.cumsum(np.bincount(gt, minlength=t.size)))))
        cs, ct, cw = canon(cols, rows, data)
        assert np.array_equal(cs, gs) and np.array_equal(ct, gt) and np.array_equal(cw, gw), (name, kind)
        if kind != "linear":  # (linear rows may hold a cell twice: clamped neighbour, weights 0 and 1)
            # rows come out ordered by column id (no host sort needed afterwards)
            same_row = np.diff(rows) == 0
            assert (np.diff(cols)[same_row] > 0).all()


def test_outer_csr_random_and_regrid_vs_oracle(hip, oracle):
    from oracle import structured

    rng = np.random.default_rng(2024)
    methods = ["mean", "harmonic_mean", "geometric_mean", "sum", "minimum", "maximum", "mode", "median", "max_overlap", "p25"]
    for i in range(12):
        ks, kt = random_raster(rng, 90), random_raster(rng, 90)
        s, t = StructuredGrid2d(Raster(**ks)), StructuredGrid2d(Raster(**kt))
        sy, sx = oracle_axes(structured, ks)
        ty, tx = oracle_axes(structured, kt)
        os_, ot, ow = structured.weights_2d("overlap", sy, sx, ty, tx)
        dcsr = s.overlap_device(t, False)
        cols, rows, data, indptr = csr_triplets(dcsr)
        cs, ct, cw = canon(cols, rows, data)
        assert np.array_equal(cs, os_) and np.array_equal(ct, ot) and np.array_equal(cw, ow)
        # end to end through the regridder API against the oracle apply on the oracle's own CSR
        src = rng.normal(size=(2,) + s.shape) + 3.0
        src[rng.random(src.shape) < 0.05] = np.nan
        method = methods[i % len(methods)]
        out = xa.OverlapRegridder(Raster(**ks), Raster(**kt), method=method).regrid(src)
        assert out.shape == (2,) + t.shape
        oindptr = oracle.to_csr_indptr(ot, t.size)
        # oracle rows in the engine's within-row order (by column id, then weight)
        expected = oracle.regrid_csr(method, src.reshape(2, -1), ow, os_, oindptr, t.size).reshape(out.shape)
        long_rows = (np.diff(oindptr) > 32).reshape(t.shape)  # cooperatively reduced rows
        short = np.broadcast_to(~long_rows, out.shape)
        if method != "geometric_mean":  # device exp/log vs libm: 1e-13 instead of bit-exact
            assert same_or_nan(out[short], expected[short]).all(), method
        np.testing.assert_allclose(out, expected, rtol=1e-13, equal_nan=True)


def test_structured_equals_polygon_path(hip):
    """test_regridder.py:295-332: the structured and the unstructured (quads) regridders give the same result."""
    rng = np.random.default_rng(11)
    ex = np.concatenate(([0.0], np.cumsum(rng.uniform(0.5, 2.0, 70))))
    ey = np.concatenate(([0.0], np.cumsum(rng.uniform(0.5, 2.0, 55))))
    src = Raster(x=0.5 * (ex[1:] + ex[:-1]), y=0.5 * (ey[1:] + ey[:-1]), dx=np.diff(ex), dy=np.diff(ey))
    tgt = Raster(x=np.arange(2.5, 80.0, 3.0), y=np.arange(70.5, -4.0, -2.5))
    data = rng.normal(size=(3, 55, 70))
    data[0, 10:14, 20:30] = np.nan
    quads_s = StructuredGrid2d(src).convert_to(xa.regrid.UnstructuredGrid2d).ugrid_topology
    quads_t = StructuredGrid2d(tgt).convert_to(xa.regrid.UnstructuredGrid2d).ugrid_topology
    for cls, kw in ((xa.OverlapRegridder, {"method": "mean"}), (xa.OverlapRegridder, {"method": "maximum"}),
                    (xa.RelativeOverlapRegridder, {}), (xa.RelativeOverlapRegridder, {"method": "conductance"})):
        a = cls(src, tgt, **kw)
        b = cls(quads_s, quads_t, **kw)
        out_a = a.regrid(data)
        out_b = b.regrid(data.reshape(3, -1)).reshape(out_a.shape)
        np.testing.assert_allclose(out_a, out_b, rtol=1e-10, atol=1e-12, equal_nan=True)
        wa, wb = a.weights_as_dataframe(), b.weights_as_dataframe()
        assert np.array_equal(wa["target_index"], wb["target_index"]) and np.array_equal(wa["source_index"], wb["source_index"])
        np.testing.assert_allclose(wa["weight"], wb["weight"], rtol=1e-10)


def test_structured_large_and_long_rows(hip, oracle):
    """~1M-cell rasters: device CSR == host outer product; coarse target rows (900 entries) take the
    block-reduced apply (1e-13 of the sequential oracle)."""
    src = Raster(x=np.arange(0.5, 1200.0), y=np.arange(899.5, 0.0, -1.0))
    tgt = Raster(x=np.arange(0.35, 1190.0, 0.7), y=np.arange(890.15, 5.0, -0.9))
    s, t = StructuredGrid2d(src), StructuredGrid2d(tgt)
    dcsr = s.overlap_device(t, False)
    hs, ht, hw = s.overlap(t, False)
    cols, rows, data, indptr = csr_triplets(dcsr)
    assert dcsr.nnz == hs.size
    assert np.array_equal(cols, hs) and np.array_equal(rows, ht) and np.array_equal(data, hw)
    rng = np.random.default_rng(3)
    field = rng.normal(size=s.shape)
    out = xa.OverlapRegridder(src, tgt, method="mean").regrid(field)
    expected = oracle.regrid_csr("mean", field.reshape(1, -1), hw, hs, indptr.astype(np.int64), t.size).reshape(t.shape)
    assert same_or_nan(out, expected).all()
    # coarse target tiling the source exactly: rows of 30 x 30 = 900 entries
    coarse = Raster(x=np.arange(15.0, 1200.0, 30.0), y=np.arange(885.0, 0.0, -30.0))
    c = StructuredGrid2d(coarse)
    assert c.shape == (30, 40)
    out = xa.OverlapRegridder(src, coarse, method="mean").regrid(field)
    hs, ht, hw = s.overlap(c, False)
    indptr = oracle.to_csr_indptr(ht, c.size)
    assert np.diff(indptr).max() == 900
    expected = oracle.regrid_csr("mean", field.reshape(1, -1), hw, hs, indptr, c.size).reshape(c.shape)
    # (means of zero-centred noise cancel: absolute floor of a few ulp of the summands)
    np.testing.assert_allclose(out, expected, rtol=1e-13, atol=1e-15)
    # conservation: the area-weighted sum of the target means is the integral of the source field
    np.testing.assert_allclose((out * c.area).sum(), field.sum(), rtol=1e-10)


def test_structured_regridders_round_trip_and_locator(hip, golden):
    g = golden("g9_structured.npz")
    for name in "bci":
        ks, kt = raster_kwargs(g, name, "src"), raster_kwargs(g, name, "tgt")
        s, t = StructuredGrid2d(Raster(**ks)), StructuredGrid2d(Raster(**kt))
        rng = np.random.default_rng(1)
        data = rng.normal(size=s.shape)
        # centroid locator: COO scatter of the golden pairs
        gs, gt, _ = golden_triplets(g, name, "locate")
        expected = np.full(t.size, np.nan)
        expected[gt] = data.ravel()[gs]
        out = xa.CentroidLocatorRegridder(Raster(**ks), Raster(**kt)).regrid(data)
        assert same_or_nan(out.ravel(), expected).all()
        # barycentric (linear) weights: rows sum to one where defined
        rg = xa.BarycentricInterpolator(Raster(**ks), Raster(**kt))
        df = rg.weights_as_dataframe()
        gs, gt, gw = golden_triplets(g, name, "linear")
        cs, ct, cw = canon(df["source_index"].to_numpy(), df["target_index"].to_numpy(), df["weight"].to_numpy())
        assert np.array_equal(cs, gs) and np.array_equal(ct, gt) and np.array_equal(cw, gw)
        out = rg.regrid(np.ones(s.shape))
        assert np.allclose(out[~np.isnan(out)], 1.0, rtol=1e-14)
        # from_weights / from_dataset round trip reproduces the result exactly
        rg1 = xa.OverlapRegridder(Raster(**ks), Raster(**kt), method="mean")
        rg2 = xa.OverlapRegridder.from_dataset(rg1.to_dataset())
        assert np.array_equal(rg1.regrid(data), rg2.regrid(data), equal_nan=True)


def test_from_outer_rejects_bad_input(hip):
    from xugrid_amd.engine import DeviceCSR

    ok = (np.array([0, 1, 2]), np.array([0, 1]), np.array([1.0, 1.0]))
    with pytest.raises(ValueError):
        DeviceCSR.from_outer((np.array([0, 2, 1]), np.array([0, 1]), np.array([1.0, 1.0])), 2, ok, 2)
    with pytest.raises(ValueError):
        DeviceCSR.from_outer(ok, 1, ok, 2)  # source index 1 outside [0, 1)
    empty = (np.array([0, 0, 0]), np.zeros(0, dtype=np.int64), np.zeros(0))
    c = DeviceCSR.from_outer(empty, 2, ok, 2)
    assert (c.n, c.m, c.nnz) == (4, 4, 0)
    assert np.isnan(c.apply(np.ones((1, 4)))).all()


ALL_METHODS = [("mean", 0), ("harmonic_mean", 1), ("geometric_mean", 2), ("sum", 3), ("minimum", 4), ("maximum", 5),
               ("mode", 6), ("p30", 7), ("first_order_conservative", 8), ("max_overlap", 9)]


def _oracle_apply(oracle, name, field, data, indices, indptr, n):
    method = ("percentile", 30.0) if name == "p30" else name
    return oracle.regrid_csr(method, field.astype(np.float64), data, indices.astype(np.int64), indptr.astype(np.int64), n)


@pytest.mark.parametrize("engine_path", ["free", "csr", None])
def test_factored_apply_vs_oracle(hip, oracle, monkeypatch, engine_path, xr_option):
    """xr_apply_outer: the matrix-free kernel walks a target cell's entries in the order of the product's CSR row,
    so it equals the sequential oracle BIT FOR BIT whatever the row length (here up to ~600 entries); the stored
    product (forced, or chosen for long x-lists) follows the CSR contract: rows <= 32 entries exact, longer 1e-13."""
    if engine_path:
        xr_option("outer_apply", engine_path)
    rng = np.random.default_rng(77)
    for case in range(8):
        ns = int(rng.integers(30, 260))
        nt = int(rng.integers(3, 180))
        ex = np.concatenate(([0.0], np.cumsum(rng.uniform(0.5, 2.0, ns))))
        ey = np.concatenate(([0.0], np.cumsum(rng.uniform(0.5, 2.0, ns // 2 + 3))))
        src = Raster(x=0.5 * (ex[1:] + ex[:-1]), y=0.5 * (ey[1:] + ey[:-1]), dx=np.diff(ex), dy=np.diff(ey))
        tgt = Raster(x=np.linspace(-3, ex[-1] + 3, nt), y=np.linspace(ey[-1] + 2, -2, max(2, nt // 2)))
        s, t = StructuredGrid2d(src), StructuredGrid2d(tgt)
        for kind in ("overlap", "linear", "locate"):
            w = {"overlap": lambda: s.overlap_device(t, case % 2 == 1), "linear": lambda: s.linear_weights_device(t),
                 "locate": lambda: s.locate_centroids_device(t)}[kind]()
            data, indices, indptr = w.download()
            long_rows = np.diff(indptr) > 32
            for K, dtype in ((1, np.float64), (2, np.float32), (3, np.float64), (9, np.float64)):
                field = (rng.normal(size=(K, s.size)) + 2.0).astype(dtype)
                field[rng.random(field.shape) < 0.05] = np.nan
                for name, mid in ALL_METHODS:
                    got = w.apply(field, mid, 30.0 if name == "p30" else 0.0)
                    exp = _oracle_apply(oracle, name, field, data, indices, indptr, t.size)
                    exact = same_or_nan(got, exp)
                    if engine_path == "free" and name not in ("mode", "p30", "geometric_mean"):
                        assert exact.all(), (case, kind, K, name)
                    elif name != "geometric_mean":
                        assert exact[:, ~long_rows].all(), (case, kind, K, name)
                    rtol = 1e-9 if name == "harmonic_mean" else 1e-12
                    np.testing.assert_allclose(got, exp, rtol=rtol, atol=1e-13, equal_nan=True)


def test_device_outer_handle(hip):
    from xugrid_amd.engine import DeviceCSR, DeviceOuter, METHOD_IDS

    ay = (np.array([0, 2, 3, 3]), np.array([0, 1, 1]), np.array([0.5, 1.5, 2.0]))
    ax = (np.array([0, 1, 3]), np.array([2, 0, 1]), np.array([1.0, 0.25, 0.75]))
    o = DeviceOuter(ay, 2, ax, 3)
    c = DeviceCSR.from_outer(ay, 2, ax, 3)
    assert (o.n, o.m, o.nnz) == (c.n, c.m, c.nnz) == (6, 6, 9)
    for a, b in zip(o.download(), c.download()):
        assert np.array_equal(a, b)
    view = o.csr()
    del o  # the borrowed view keeps the owner alive
    src = np.array([[1.0, 2.0, 4.0, 8.0, np.nan, 32.0]])
    assert same_or_nan(view.apply(src, METHOD_IDS["sum"]), c.apply(src, METHOD_IDS["sum"])).all()
    o = DeviceOuter(ay, 2, ax, 3)
    out = o.apply(src, METHOD_IDS["sum"])
    # row 0 = y-list {0, 1} x x-list {2}: the (unweighted, reduce.py:66-67) sum of cells 2 and 5; rows 4, 5: empty y-list -> NaN
    assert out[0, 0] == 4.0 + 32.0 and np.isnan(out[0, 4:]).all()
    assert same_or_nan(out, c.apply(src, METHOD_IDS["sum"])).all()
    with pytest.raises(ValueError):
        o.apply(np.ones((1, 5)))
    with pytest.raises(ValueError):
        DeviceOuter(ay, 1, ax, 3)  # source index 1 outside [0, 1)
    empty = (np.array([0, 0, 0]), np.zeros(0, dtype=np.int64), np.zeros(0))
    e = DeviceOuter(empty, 2, ax, 3)
    assert (e.n, e.nnz) == (4, 0) and np.isnan(e.apply(np.ones((2, 6)))).all()
    assert e.download()[2].tolist() == [0, 0, 0, 0, 0]


def test_factored_apply_large_and_anisotropic(hip, oracle, monkeypatch, xr_option):
    """2000 x 1500 source cells: coarsening by 40 along y only keeps x-lists short -> matrix-free with 80-entry rows,
    bit-identical to the sequential oracle; host chunking of many variables; float32 sources."""
    ns_x, ns_y = 2000, 1500
    src = Raster(x=np.arange(0.5, ns_x), y=np.arange(ns_y - 0.5, 0.0, -1.0))
    tgt = Raster(x=np.arange(0.85, ns_x - 1.0, 0.9), y=np.arange(ns_y - 20.0, 0.0, -40.0))
    s, t = StructuredGrid2d(src), StructuredGrid2d(tgt)
    w = s.overlap_device(t, False)
    data, indices, indptr = w.download()
    assert np.diff(indptr).max() >= 80
    rng = np.random.default_rng(5)
    field = rng.normal(size=(5, s.size)).astype(np.float32)
    field[rng.random(field.shape) < 0.01] = np.nan
    for name in ("mean", "sum", "maximum", "max_overlap"):
        got = w.apply(field, dict(ALL_METHODS)[name])
        exp = _oracle_apply(oracle, name, field, data, indices, indptr, t.size)
        assert same_or_nan(got, exp).all(), name
    out = xa.OverlapRegridder(src, tgt, method="mean").regrid(field.reshape(5, ns_y, ns_x))
    exp = _oracle_apply(oracle, "mean", field, data, indices, indptr, t.size)
    assert same_or_nan(out.reshape(5, -1), exp).all()
    # host buffers larger than the device staging budget go through in chunks of variables (test hook)
    xr_option("apply_chunk_bytes", str(2 * (s.size * 4 + t.size * 8) + 1))
    assert same_or_nan(w.apply(field, 0), exp).all()
