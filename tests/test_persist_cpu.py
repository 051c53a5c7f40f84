"""
Weights persistence in the reference's dataset layout (xugrid/regrid/regridder.py:264-271,334-361,
regrid/unstructured.py:217-220, regrid/structured.py:436-450,603-608, ugrid/conventions.py:158-177).  xarray is absent
in this image, so the layout is pinned by TRANSCRIPTION: the variable names, dims and attrs asserted below are the ones
those lines produce.  Host logic only -- no device call.
"""
import numpy as np
import pytest

import xugrid_amd as xa
from xugrid_amd import meshgen
from xugrid_amd.regrid import persist
from xugrid_amd.regrid.regridder import (BarycentricInterpolator, CentroidLocatorRegridder, OverlapRegridder,
                                         RelativeOverlapRegridder, setup_grid)
from xugrid_amd.regrid.structured import Raster
from xugrid_amd.sparse import MatrixCOO, MatrixCSR


def _fake_regridder(cls, source, target, seed=0):
    """A regridder with host weights only (what ``from_weights`` leaves before the first apply)."""
    rng = np.random.default_rng(seed)
    r = cls.__new__(cls)
    r._source, r._target = setup_grid(source), setup_grid(target)
    r._device_weights = None
    n, m = r._target.size, r._source.size
    if cls is CentroidLocatorRegridder:
        row = np.sort(rng.choice(n, size=n // 2, replace=False))
        r._weights = MatrixCOO(np.ones(row.size), row, rng.integers(0, m, row.size), n, m, row.size)
    else:
        counts = rng.integers(0, 4, n)
        indptr = np.concatenate([[0], np.cumsum(counts)])
        r._weights = MatrixCSR(rng.random(indptr[-1]), rng.integers(0, m, indptr[-1]), indptr, n, m, int(indptr[-1]))
        r._setup_regrid(next(iter(cls._METHODS)))
    return r


@pytest.fixture
def mesh():
    xy, faces = meshgen.triangle_mesh(60, 3)
    return xa.Ugrid2d(xy[:, 0], xy[:, 1], -1, faces)


def test_unstructured_layout_is_the_references(mesh):
    r = _fake_regridder(OverlapRegridder, mesh, mesh)
    ds = r.to_reference_dataset()
    # regridder.py:264-271: xr.Dataset({"__regrid_<field>": value}) -- 1-D arrays get a dim named after the variable
    for field in ("data", "indices", "indptr"):
        assert ds[f"__regrid_{field}"].dims == (f"__regrid_{field}",)
    for field in ("n", "m", "nnz"):
        assert ds[f"__regrid_{field}"].dims == () and ds[f"__regrid_{field}"].item() == getattr(r._weights, field)
    for name in ("__source", "__target"):
        # unstructured.py:217-220
        marker = ds[name + "_type"]
        assert marker.dims == () and marker.item() == -1 and marker.attrs == {"type": "UnstructuredGrid2d"}
        # conventions.py:158-177 after Ugrid2d.rename(name)
        topo = ds[name]
        assert topo.item() == 0 and topo.attrs["cf_role"] == "mesh_topology" and topo.attrs["topology_dimension"] == 2
        assert topo.attrs["node_coordinates"] == f"{name}_node_x {name}_node_y"
        assert topo.attrs["face_node_connectivity"] == f"{name}_face_nodes"
        assert topo.attrs["face_dimension"] == f"{name}_nFaces" and topo.attrs["node_dimension"] == f"{name}_nNodes"
        faces = ds[f"{name}_face_nodes"]
        assert faces.dims == (f"{name}_nFaces", f"{name}_nMax_face_nodes")
        assert faces.attrs["cf_role"] == "face_node_connectivity" and faces.attrs["start_index"] == 0
        assert ds[f"{name}_node_x"].dims == (f"{name}_nNodes",) and ds[f"{name}_node_y"].dims == (f"{name}_nNodes",)
        assert np.array_equal(faces.data, mesh.face_node_connectivity)
    assert ds.attrs["Conventions"] == "CF-1.9 UGRID-1.0"


def test_structured_layout_is_the_references(mesh):
    # descending y, non-equidistant x with explicit bounds
    xb = np.column_stack([[0.0, 1.0, 3.0, 3.5], [1.0, 3.0, 3.5, 6.0]])
    raster = Raster(xb.mean(axis=1), np.array([2.5, 1.5, 0.5]), xbounds=xb)
    r = _fake_regridder(RelativeOverlapRegridder, raster, mesh)
    ds = r.to_reference_dataset()
    assert ds["__source_type"].attrs == {"type": "StructuredGrid2d"} and ds["__source_type"].item() == -1
    # structured.py:436-450: export_name = name + "_" + axis name; midpoints ASCENDING, bounds (n, 2), nbounds = [0, 1]
    assert ds["__source_x"].dims == ("__source_x",) and np.array_equal(ds["__source_x"].data, xb.mean(axis=1))
    assert ds["__source_y"].dims == ("__source_y",) and np.array_equal(ds["__source_y"].data, [0.5, 1.5, 2.5])
    assert ds["__source_xbounds"].dims == ("__source_x", "__source_xnbounds")
    assert np.array_equal(ds["__source_xbounds"].data, xb)
    assert np.array_equal(ds["__source_ybounds"].data, [[0.0, 1.0], [1.0, 2.0], [2.0, 3.0]])
    assert np.array_equal(ds["__source_xnbounds"].data, [0, 1]) and np.array_equal(ds["__source_ynbounds"].data, [0, 1])
    # the data variable <name>: NaN on the dims of the FIRST merged axis (x, structured.py:604-606)
    assert ds["__source"].dims == ("__source_x", "__source_xnbounds") and np.isnan(ds["__source"].data).all()
    assert {"__source_x", "__source_xbounds", "__source_y", "__source_ybounds"} <= set(ds.coords)


@pytest.mark.parametrize("cls", [OverlapRegridder, RelativeOverlapRegridder, BarycentricInterpolator,
                                 CentroidLocatorRegridder])
@pytest.mark.parametrize("structured_source", [False, True])
def test_reference_layout_round_trip(cls, structured_source, mesh, tmp_path):
    raster = Raster(np.arange(6.0) + 0.5, np.arange(4.0)[::-1] + 0.5)
    source = raster if structured_source else mesh
    r = _fake_regridder(cls, source, mesh if structured_source else raster, seed=4)
    ds = r.to_reference_dataset()
    path = tmp_path / "weights.npz"
    r.to_file(path, layout="reference")
    for back in (cls.from_reference_dataset(ds), cls.from_dataset(ds), cls.from_weights(ds, r._target),
                 cls.from_file(path)):
        assert type(back._weights) is type(r._weights)
        for a, b in zip(back._weights, r._weights):
            assert np.array_equal(np.asarray(a), np.asarray(b))
        assert back._source.shape == r._source.shape and back._target.shape == r._target.shape
        assert type(back._source) is type(r._source) and type(back._target) is type(r._target)
        if not structured_source:
            g, h = back._source.ugrid_topology, r._source.ugrid_topology
            assert np.array_equal(g.node_x, h.node_x) and np.array_equal(g.node_y, h.node_y)
            assert np.array_equal(g.face_node_connectivity, h.face_node_connectivity)
        # a second generation writes the same variables (names do not grow a prefix per round trip)
        again = back.to_reference_dataset()
        assert set(again) == set(ds)
        for k in ds:
            assert again[k].dims == ds[k].dims and np.array_equal(again[k].data, ds[k].data, equal_nan=True), k


class _Var:
    """What an xr.DataArray offers the reader: attrs, encoding, to_numpy."""

    def __init__(self, data, attrs=None, encoding=None):
        self._data, self.attrs, self.encoding = np.asarray(data), dict(attrs or {}), dict(encoding or {})

    def to_numpy(self):
        return self._data


def test_reads_a_dataset_as_xugrid_writes_it():
    """A dataset with the quirks a netCDF round trip of xugrid's output can carry: 1-based connectivity
    (``start_index`` 1), a ``_FillValue`` moved to ``encoding``, float connectivity with NaN fill, an extra
    ``edge_nodes`` variable, a mixed triangle / quad mesh."""
    node_x = np.array([0.0, 1.0, 2.0, 0.0, 1.0, 2.0])
    node_y = np.array([0.0, 0.0, 0.0, 1.0, 1.0, 1.0])
    faces0 = np.array([[0, 1, 4, 3], [1, 2, 5, -1], [1, 5, 4, -1]])
    one_based = np.where(faces0 < 0, -999, faces0 + 1)
    topo = {"cf_role": "mesh_topology", "topology_dimension": 2, "node_coordinates": "__source_node_x __source_node_y",
            "face_node_connectivity": "__source_face_nodes", "edge_node_connectivity": "__source_edge_nodes"}
    base = {
        "__regrid_data": _Var([0.5, 0.5, 1.0]), "__regrid_indices": _Var([0, 1, 2]), "__regrid_indptr": _Var([0, 2, 3]),
        "__regrid_n": _Var(2), "__regrid_m": _Var(3), "__regrid_nnz": _Var(3),
        "__source": _Var(0, topo), "__source_node_x": _Var(node_x), "__source_node_y": _Var(node_y),
        "__source_edge_nodes": _Var(np.zeros((7, 2), dtype=int)),
        "__source_type": _Var(-1, {"type": "UnstructuredGrid2d"}),
        "__target_type": _Var(-1, {"type": "StructuredGrid2d"}),
        "__target_x": _Var([0.5, 1.5]), "__target_xbounds": _Var([[0.0, 1.0], [1.0, 2.0]]),
        "__target_y": _Var([0.5]), "__target_ybounds": _Var([[0.0, 1.0]]),
    }
    variants = {
        "attrs fill": _Var(one_based, {"start_index": 1, "_FillValue": -999}),
        "encoding fill": _Var(one_based, {"start_index": 1}, {"_FillValue": -999}),
        "float NaN fill": _Var(np.where(faces0 < 0, np.nan, faces0 + 1.0), {"start_index": 1}, {"_FillValue": np.nan}),
        "zero based": _Var(faces0, {"start_index": 0, "_FillValue": -1}),
    }
    for label, var in variants.items():
        ds = dict(base, __source_face_nodes=var)
        assert persist.is_reference_layout(ds)
        r = OverlapRegridder.from_dataset(ds)
        grid = r._source.ugrid_topology
        assert np.array_equal(grid.face_node_connectivity, faces0), label
        assert np.array_equal(grid.node_x, node_x) and r._source.shape == (3,)
        assert r._target.shape == (1, 2) and r._weights.nnz == 3 and r._method.name == "mean"
    with pytest.raises(KeyError):
        OverlapRegridder.from_dataset({k: v for k, v in base.items() if k != "__source_node_x"}
                                      | {"__source_face_nodes": variants["zero based"]})


def test_flat_layout_still_reads(mesh, tmp_path):
    r = _fake_regridder(OverlapRegridder, mesh, Raster(np.arange(3.0) + 0.5, np.arange(2.0) + 0.5))
    flat = r.to_dataset()
    assert not persist.is_reference_layout(flat)
    back = OverlapRegridder.from_dataset(flat)
    assert np.array_equal(back._weights.data, r._weights.data) and back._target.shape == (2, 3)
    r.to_file(tmp_path / "w.npz")
    assert np.array_equal(OverlapRegridder.from_file(tmp_path / "w.npz")._weights.indices, r._weights.indices)
    with pytest.raises(ValueError):
        r.to_file(tmp_path / "w2.npz", layout="netcdf")


class _DA:
    def __init__(self, values, dims, grid=None):
        self.values, self.dims = values, dims
        if grid is not None:
            self.grid = grid


def test_source_dims_come_from_the_data(mesh):
    """regridder.py:231-251: after from_dataset the source grid is called "__source"; the dims to regrid over are the
    DATA's (``("y", "x")`` / the data grid's core dimension), not ``__source_nFaces``."""
    r = _fake_regridder(OverlapRegridder, mesh, mesh)
    back = OverlapRegridder.from_reference_dataset(r.to_reference_dataset())
    assert back._source.dims == ("__source_nFaces",)
    seen = {}
    back._regrid_array = lambda a: seen.setdefault("shape", a.shape)
    n = mesh.n_face
    back.regrid(_DA(np.zeros((n, 4)), ("mesh2d_nFaces", "time"), grid=mesh))  # UgridDataArray-like: core dim of ITS grid
    assert seen.pop("shape") == (4, n)
    back.regrid(_DA(np.zeros((4, n)), ("time", "whatever_nFaces")))  # bare DataArray-like: the last dim
    assert seen.pop("shape") == (4, n)
    back.regrid(_DA(np.zeros((n, 4)), ("__source_nFaces", "time")))  # names the regridder's own dim
    assert seen.pop("shape") == (4, n)
    with pytest.raises(ValueError, match="source dimensions"):
        back.regrid(_DA(np.zeros((4, n)), ("time", "face"), grid=mesh))
    # structured source reloaded from the reference layout: dims are ("y", "x") whatever the stored names
    raster = Raster(np.arange(5.0) + 0.5, np.arange(3.0) + 0.5)
    rs = _fake_regridder(OverlapRegridder, raster, mesh)
    back = OverlapRegridder.from_reference_dataset(rs.to_reference_dataset())
    assert back._source.dims == ("__source_y", "__source_x")
    back._regrid_array = lambda a: seen.setdefault("shape", a.shape)
    back.regrid(_DA(np.zeros((5, 2, 3)), ("x", "layer", "y")))
    assert seen.pop("shape") == (2, 3, 5)
    with pytest.raises(ValueError, match="source dimensions"):
        back.regrid(_DA(np.zeros((3, 5)), ("lat", "lon")))
