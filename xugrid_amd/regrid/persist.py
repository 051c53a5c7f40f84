"""
Weights persistence in the REFERENCE's dataset layout -- what ``BaseRegridder.to_dataset`` writes
and ``from_weights`` / ``from_dataset`` read in xugrid (xugrid/regrid/regridder.py:264-271, :334-361):

  weights      ``__regrid_data / __regrid_indices / __regrid_indptr`` (CSR) or ``__regrid_row / __regrid_col`` (COO),
               each 1-D with a dim named after the variable (what ``xr.Dataset({name: ndarray})`` does), and the
               scalars ``__regrid_n / __regrid_m / __regrid_nnz``
  grid marker  ``__source_type`` / ``__target_type``: scalar -1 whose ``attrs["type"]`` is "UnstructuredGrid2d" or
               "StructuredGrid2d" (regrid/unstructured.py:217-220, regrid/structured.py:603-608)
  unstructured ``Ugrid2d.rename(name).to_dataset()`` (ugrid/ugrid2d.py:350-421, ugrid/conventions.py:158-177): the
               topology variable ``<name>`` (0, UGRID attrs), ``<name>_node_x / _node_y`` on ``<name>_nNodes``,
               ``<name>_face_nodes`` on ``(<name>_nFaces, <name>_nMax_face_nodes)`` with ``start_index`` /
               ``_FillValue`` attrs
  structured   per axis (regrid/structured.py:436-450): coordinate ``<name>_x`` = ASCENDING midpoints,
               ``<name>_xbounds`` on ``(<name>_x, <name>_xnbounds)`` = the ascending bounds, ``<name>_xnbounds`` =
               [0, 1]; the same for y; the data variable ``<name>`` is a NaN block on the x dims

xarray is optional and absent in this image, so the layout is carried by ``RefDataset`` -- a dict of
``RefVariable(dims, data, attrs)``, the three things an ``xr.DataArray`` is made of.  Reading is duck-typed
(``ds[name]`` must offer ``attrs`` and convert with ``np.asarray``), so a real ``xr.Dataset`` written by xugrid is
accepted as it is, and ``RefDataset.to_xarray()`` gives one back when xarray is importable.
"""
import json

import numpy as np

from ..sparse import MatrixCOO, MatrixCSR
from ..ugrid2d import Ugrid2d

FILL_VALUE = -1


class RefVariable:
    """dims + data + attrs: the part of an ``xr.DataArray`` the reference's persistence code reads."""

    __slots__ = ("dims", "data", "attrs")

    def __init__(self, dims, data, attrs=None):
        self.data = np.asarray(data)
        self.dims = tuple(dims)
        if len(self.dims) != self.data.ndim:
            raise ValueError(f"{len(self.dims)} dims for {self.data.ndim}-dimensional data")
        self.attrs = dict(attrs or {})

    def __array__(self, dtype=None, copy=None):
        return self.data if dtype is None else self.data.astype(dtype)

    @property
    def values(self):
        return self.data

    def to_numpy(self):
        return self.data

    def item(self):
        return self.data.item()


class RefDataset(dict):
    """name -> RefVariable; ``coords`` names the variables xarray would hold as coordinates."""

    def __init__(self, *args, attrs=None, coords=(), **kwargs):
        super().__init__(*args, **kwargs)
        self.attrs = dict(attrs or {})
        self.coord_names = set(coords)

    @property
    def coords(self):
        return {k: self[k] for k in self.coord_names if k in self}

    def merge(self, other: "RefDataset") -> "RefDataset":
        """``xr.merge(..., compat="override")``: the first definition of a name wins."""
        for k, v in other.items():
            self.setdefault(k, v)
        self.coord_names |= other.coord_names
        for k, v in other.attrs.items():
            self.attrs.setdefault(k, v)
        return self

    # ---- file form: one .npz (arrays) + a JSON sidecar variable holding dims / attrs
    def save(self, path) -> None:
        meta = {
            "attrs": self.attrs,
            "coords": sorted(self.coord_names),
            "variables": {k: {"dims": list(v.dims), "attrs": _jsonable(v.attrs)} for k, v in self.items()},
        }
        np.savez_compressed(path, __meta__=np.array(json.dumps(meta)), **{k: v.data for k, v in self.items()})

    @staticmethod
    def load(path) -> "RefDataset":
        with np.load(path, allow_pickle=False) as archive:
            meta = json.loads(str(archive["__meta__"].item()))
            ds = RefDataset(attrs=meta["attrs"], coords=meta["coords"])
            for k, m in meta["variables"].items():
                ds[k] = RefVariable(m["dims"], archive[k], m["attrs"])
        return ds

    def to_xarray(self):
        import xarray as xr  # optional dependency

        coords = {k: (v.dims, v.data, v.attrs) for k, v in self.items() if k in self.coord_names}
        data = {k: (v.dims, v.data, v.attrs) for k, v in self.items() if k not in self.coord_names}
        return xr.Dataset(data, coords=coords, attrs=self.attrs)


def _jsonable(attrs):
    return {k: (v.item() if isinstance(v, np.generic) else v) for k, v in attrs.items()}


# ---------------------------------------------------------------------------------------------- writing
def weights_to_reference(matrix) -> RefDataset:
    ds = RefDataset()
    for field, value in zip(matrix._fields, matrix):
        name = f"__regrid_{field}"
        value = np.asarray(value)
        ds[name] = RefVariable((name,) if value.ndim == 1 else (), value)
    return ds


def ugrid2d_to_reference(grid: Ugrid2d, name: str) -> RefDataset:
    """``grid.rename(name).to_dataset()`` + the ``<name>_type`` marker."""
    faces = np.asarray(grid.face_node_connectivity)
    topology_attrs = {
        "cf_role": "mesh_topology",
        "long_name": "Topology data of 2D mesh",
        "topology_dimension": 2,
        "node_dimension": f"{name}_nNodes",
        "face_dimension": f"{name}_nFaces",
        "max_face_nodes_dimension": f"{name}_nMax_face_nodes",
        "face_node_connectivity": f"{name}_face_nodes",
        "node_coordinates": f"{name}_node_x {name}_node_y",
    }
    ds = RefDataset(attrs={"Conventions": "CF-1.9 UGRID-1.0"}, coords=(f"{name}_node_x", f"{name}_node_y"))
    ds[name] = RefVariable((), 0, topology_attrs)
    ds[f"{name}_face_nodes"] = RefVariable(
        (f"{name}_nFaces", f"{name}_nMax_face_nodes"), faces,
        {"cf_role": "face_node_connectivity", "start_index": 0, "_FillValue": FILL_VALUE},
    )
    ds[f"{name}_node_x"] = RefVariable((f"{name}_nNodes",), grid.node_x, {"standard_name": "projection_x_coordinate"})
    ds[f"{name}_node_y"] = RefVariable((f"{name}_nNodes",), grid.node_y, {"standard_name": "projection_y_coordinate"})
    ds[f"{name}_type"] = RefVariable((), -1, {"type": "UnstructuredGrid2d"})
    return ds


def structured2d_to_reference(grid, name: str) -> RefDataset:
    """StructuredGrid2d.to_dataset (structured.py:603-608) = merge of the two StructuredGrid1d.to_dataset (:436-450)."""
    ds = RefDataset()
    for axis in (grid.xbounds, grid.ybounds):  # merge order of :604-606: x first, so <name> lies on the x dims
        export = f"{name}_{_axis_letter(axis.name)}"
        part = RefDataset(coords=(export, export + "bounds", export + "nbounds"))
        part[name] = RefVariable((export, export + "nbounds"), np.full((axis.size, 2), np.nan))
        part[export] = RefVariable((export,), axis.midpoints)
        part[export + "bounds"] = RefVariable((export, export + "nbounds"), axis.bounds)
        part[export + "nbounds"] = RefVariable((export + "nbounds",), np.arange(2))
        ds.merge(part)
    ds[f"{name}_type"] = RefVariable((), -1, {"type": "StructuredGrid2d"})
    return ds


def _axis_letter(axis_name: str) -> str:
    # the reference appends the axis' own name ("x" / "y"; after a reload "__source_x" -> the last letter keeps the
    # exported names stable instead of growing a prefix per round trip)
    return axis_name[-1]


def grid_to_reference(grid, name: str) -> RefDataset:
    from .structured import StructuredGrid2d
    from .unstructured import UnstructuredGrid2d

    if isinstance(grid, UnstructuredGrid2d):
        return ugrid2d_to_reference(grid.ugrid_topology, name)
    if isinstance(grid, StructuredGrid2d):
        return structured2d_to_reference(grid, name)
    raise TypeError(f"cannot write a {type(grid).__name__}")


# ---------------------------------------------------------------------------------------------- reading
def _attrs(ds, name):
    var = ds[name]
    attrs = getattr(var, "attrs", None)
    if attrs is None:
        raise TypeError(f"variable {name} carries no attrs: not a dataset in the reference layout")
    return attrs


def _np(ds, name):
    var = ds[name]
    return np.asarray(var.to_numpy() if hasattr(var, "to_numpy") else var)


def grid_kind(ds, name: str) -> str:
    """regridder.py:340,355: ``ds[name + "_type"].attrs["type"]``."""
    return str(_attrs(ds, name + "_type")["type"])


def ugrid2d_from_reference(ds, name: str) -> Ugrid2d:
    """Ugrid2d.from_dataset(ds, name) (ugrid2d.py:246-348): names through the topology variable's UGRID attrs,
    ``start_index`` and ``_FillValue`` honoured."""
    topo = _attrs(ds, name)
    x_name, y_name = str(topo["node_coordinates"]).split()[:2]
    faces_name = str(topo["face_node_connectivity"])
    fattrs = dict(_attrs(ds, faces_name))
    encoding = getattr(ds[faces_name], "encoding", {}) or {}
    fill = encoding.get("_FillValue", fattrs.get("_FillValue", FILL_VALUE))
    start_index = int(fattrs.get("start_index", 0))
    raw = _np(ds, faces_name)
    invalid = np.isnan(raw) if raw.dtype.kind == "f" else np.zeros(raw.shape, dtype=bool)
    if fill is not None and not (isinstance(fill, float) and np.isnan(fill)):
        invalid |= raw == fill
    faces = np.where(invalid, 0, raw).astype(np.int64) - start_index
    faces[invalid] = FILL_VALUE
    return Ugrid2d(_np(ds, x_name).astype(np.float64), _np(ds, y_name).astype(np.float64), FILL_VALUE, faces, name=name)


def structured2d_from_reference(ds, name: str):
    """``setup_grid(weights, name_x="__source_x", name_y="__source_y")`` (regridder.py:344-346): midpoints from the
    index ``<name>_x``, bounds from the coordinate ``<name>_xbounds`` (structured.py:33-53)."""
    from .structured import Raster, StructuredGrid2d

    x, y = _np(ds, f"{name}_x"), _np(ds, f"{name}_y")
    return StructuredGrid2d(
        Raster(x, y, xbounds=_np(ds, f"{name}_xbounds"), ybounds=_np(ds, f"{name}_ybounds")),
        name_x=f"{name}_x", name_y=f"{name}_y",
    )


def grid_from_reference(ds, name: str):
    from .unstructured import UnstructuredGrid2d

    if grid_kind(ds, name) == "UnstructuredGrid2d":
        return UnstructuredGrid2d(ugrid2d_from_reference(ds, name))
    return structured2d_from_reference(ds, name)


def csr_from_reference(ds) -> MatrixCSR:
    return MatrixCSR(
        _np(ds, "__regrid_data"), _np(ds, "__regrid_indices"), _np(ds, "__regrid_indptr"),
        int(_np(ds, "__regrid_n").item()), int(_np(ds, "__regrid_m").item()), int(_np(ds, "__regrid_nnz").item()),
    )


def coo_from_reference(ds) -> MatrixCOO:
    return MatrixCOO(
        _np(ds, "__regrid_data"), _np(ds, "__regrid_row"), _np(ds, "__regrid_col"),
        int(_np(ds, "__regrid_n").item()), int(_np(ds, "__regrid_m").item()), int(_np(ds, "__regrid_nnz").item()),
    )


def is_reference_layout(ds) -> bool:
    """True when ``ds["__source_type"]`` is an attrs-typed marker (the reference's layout) rather than the flat
    dict of this package's ``to_dataset``."""
    try:
        return "type" in getattr(ds["__source_type"], "attrs", {})
    except (KeyError, TypeError):
        return False
