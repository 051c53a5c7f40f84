"""
Rectilinear grids on the regridder boundary: the part of xugrid/regrid/structured.py the
unstructured hot path needs -- bounds inference of ``StructuredGrid1d``
(:33-83), ``directional_bounds`` (:110-116), ``coords``/``shape``/``dims`` (:85-108, :454-483) and
``StructuredGrid2d.convert_to(UnstructuredGrid2d)`` (:489-501).  The separable structured ->
structured fast paths (:503-601) are SURVEY 8(f) rank 1 ("next"); until they exist a structured
pair is promoted to quads and goes through the polygon clip like everything else.

xarray is optional (absent in this image): a raster is described by ``Raster`` -- the coordinates
an ``xr.DataArray`` would carry -- or by a real DataArray/Dataset when xarray is importable.
"""
from typing import Optional

import numpy as np

from ..ugrid2d import Ugrid2d
from .unstructured import UnstructuredGrid2d


class Raster:
    """
    Coordinates of a rectilinear raster with dims ``("y", "x")``: midpoints plus either explicit
    bounds (``xbounds``/``ybounds``, shape (n, 2)), cell sizes (``dx``/``dy``, scalar or (n,);
    sign ignored) or nothing (equidistant midpoints).  Mirrors what StructuredGrid1d reads from a
    DataArray (structured.py:33-83).
    """

    def __init__(self, x, y, dx=None, dy=None, xbounds=None, ybounds=None, name=None):
        self.x = np.asarray(x, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)
        self.dx, self.dy = dx, dy
        self.xbounds, self.ybounds = xbounds, ybounds
        self.name = name

    def axis(self, name):
        if name.endswith("x"):
            return self.x, self.dx, self.xbounds
        return self.y, self.dy, self.ybounds


def _axis_from_obj(obj, name):
    if isinstance(obj, Raster):
        return obj.axis(name)
    # duck-typed xarray object
    index = np.asarray(obj[name].values if hasattr(obj[name], "values") else obj[name], dtype=np.float64)
    coords = getattr(obj, "coords", {})
    bounds = np.asarray(coords[f"{name}bounds"]) if f"{name}bounds" in coords else None
    size = np.asarray(coords[f"d{name}"]) if f"d{name}" in coords else None
    return index, size, bounds


class StructuredGrid1d:
    def __init__(self, obj, name: str):
        index, size, bounds = _axis_from_obj(obj, name)
        if index.ndim != 1 or index.size == 0:
            raise ValueError(f"{name} must be a non-empty 1-D coordinate")
        d = np.diff(index)
        if index.size > 1 and (d < 0).all():
            midpoints, flipped, side = index[::-1], True, "right"
        elif index.size == 1 or (d > 0).all():
            midpoints, flipped, side = index, False, "left"
        else:
            raise ValueError(f"{name} is not monotonic for array {getattr(obj, 'name', None)}")
        if bounds is not None:
            bounds = np.asarray(bounds, dtype=np.float64)
            size_value = bounds[:, 1] - bounds[:, 0]
        else:
            if size is not None:
                size_value = np.asarray(size, dtype=np.float64)
            else:
                if index.size < 2:
                    raise ValueError(f"cannot infer the cell size of a single-cell axis {name}")
                size_value = np.diff(midpoints)
                atolx = 1.0e-4 * size_value[0]
                if not np.allclose(size_value, size_value[0], atolx):
                    raise ValueError(
                        f"DataArray has to be equidistant along {name}, or explicit bounds must be given as "
                        f'"{name}bounds", or cellsizes must be as "d{name}"'
                    )
                size_value = np.full_like(midpoints, size_value[0])
            abs_size = np.abs(size_value)
            start = midpoints - 0.5 * abs_size
            end = midpoints + 0.5 * abs_size
            bounds = np.column_stack((start, end))
        self.name = name
        self.midpoints = midpoints
        self.bounds = bounds
        self.flipped = flipped
        self.side = side
        self.dname = f"d{name}"
        self.dvalue = size_value
        self.index = index

    @property
    def coords(self) -> dict:
        coords = {self.name: self.index}
        coords[self.dname] = self.dvalue
        return coords

    @property
    def ndim(self):
        return 1

    @property
    def dims(self):
        return (self.name,)

    @property
    def size(self):
        return len(self.bounds)

    @property
    def length(self):
        return np.abs(self.bounds[:, 1] - self.bounds[:, 0])

    @property
    def directional_bounds(self):
        return self.bounds[::-1, :].copy() if self.flipped else self.bounds


class StructuredGrid2d:
    """Raster topology; face id of cell (iy, ix) = iy * nx + ix in the raster's own order."""

    def __init__(self, obj, name_x: str = "x", name_y: str = "y"):
        self.xbounds = StructuredGrid1d(obj, name_x)
        self.ybounds = StructuredGrid1d(obj, name_y)
        self._unstructured: Optional[UnstructuredGrid2d] = None

    @property
    def coords(self) -> dict:
        return {**self.ybounds.coords, **self.xbounds.coords}

    @property
    def ndim(self):
        return 2

    @property
    def dims(self):
        return self.ybounds.dims + self.xbounds.dims

    @property
    def size(self):
        return self.ybounds.size * self.xbounds.size

    @property
    def shape(self):
        return (self.ybounds.size, self.xbounds.size)

    @property
    def area(self):
        return np.multiply.outer(self.ybounds.length, self.xbounds.length)

    def convert_to(self, matched_type):
        if matched_type == StructuredGrid2d:
            return self
        if matched_type == UnstructuredGrid2d:
            if self._unstructured is None:
                ugrid2d = Ugrid2d.from_structured_bounds(
                    self.xbounds.directional_bounds, self.ybounds.directional_bounds
                )
                self._unstructured = UnstructuredGrid2d(ugrid2d)
            return self._unstructured
        raise TypeError(f"Cannot convert StructuredGrid2d to {matched_type.__name__}")

    def to_dataset(self, name: str):
        return {
            name + "_x": self.xbounds.index,
            name + "_y": self.ybounds.index,
            name + "_xbounds": self.xbounds.directional_bounds,
            name + "_ybounds": self.ybounds.directional_bounds,
            name + "_type": "StructuredGrid2d",
        }

    @staticmethod
    def from_dataset(dataset, name: str):
        xb = np.asarray(dataset[name + "_xbounds"])
        yb = np.asarray(dataset[name + "_ybounds"])
        return StructuredGrid2d(
            Raster(np.asarray(dataset[name + "_x"]), np.asarray(dataset[name + "_y"]), xbounds=_undirect(xb),
                   ybounds=_undirect(yb))
        )


def _undirect(bounds):
    """directional bounds -> ascending-midpoint order expected by StructuredGrid1d."""
    if bounds.shape[0] > 1 and bounds[0, 0] > bounds[-1, 0]:
        return bounds[::-1].copy()
    return bounds
