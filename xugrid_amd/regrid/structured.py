"""
Rectilinear grids on the regridder boundary -- counterpart of xugrid/regrid/structured.py:

  * bounds inference of ``StructuredGrid1d`` (:33-83), ``directional_bounds`` (:110-116),
    ``coords``/``shape``/``dims`` (:85-108, :454-483) and ``StructuredGrid2d.convert_to(UnstructuredGrid2d)``
    (:489-501), which is how a raster enters the polygon hot path when the other grid is unstructured;
  * the separable structured -> structured constructions: per axis ``overlap`` (:335-356,
    through regrid/overlap_1d.py:162-257), ``locate_centroids`` (:358-377, :118-156) and ``linear_weights``
    (:379-403, :201-315) are O(n) host numpy exactly as in the reference; what is O(n_y * n_x) -- the outer
    product of the two axes and its ordering by target cell (``broadcast_sorted`` :503-531, regrid/utils.py:17-36)
    -- is assembled straight into a device CSR by ``xr_csr_from_outer`` (no argsort: the CSR offsets of an
    outer product have a closed form).

xarray is optional (absent in this image): a raster is described by ``Raster`` -- the coordinates
an ``xr.DataArray`` would carry -- or by a real DataArray/Dataset when xarray is importable.
"""
from typing import Optional

import numpy as np

from .. import engine
from ..ugrid2d import Ugrid2d
from .unstructured import UnstructuredGrid2d

IntDType = engine.IntDType


def _clipped_search(a, v, side, add):
    """
    overlap_1d.py:77-101 for one pair of 1-D arrays: a binary search of ``v`` in the NaN-free part of
    ``a`` that never looks at its last element (``n = jj - 1``), mapped back to positions of ``a`` and
    shifted by ``add``, clipped to [0, a.size]; -1 where ``v`` is NaN.
    """
    keep = np.flatnonzero(~np.isnan(a))
    if keep.size == 0:
        raise ValueError("bounds contain no valid values")
    packed = a[keep]
    pos = np.searchsorted(packed[:-1], v, side=side)
    out = np.clip(keep[pos] + add, 0, a.size)
    out[np.isnan(v)] = -1
    return out


def _overlap_1d(source_bounds, target_bounds):
    """
    overlap_1d.py:246-257: all (source, target) interval pairs with a positive overlap length, target-major,
    source ascending.  Candidates per target are the sources from the one containing its lower bound to the
    one containing its upper bound (binary searches as in the reference, :213-218).
    """
    sl, su = source_bounds[:, 0], source_bounds[:, 1]
    tl, tu = target_bounds[:, 0], target_bounds[:, 1]
    first = _clipped_search(sl, tl, "right", -1)
    last = _clipped_search(su, tu, "left", 1)
    count = np.maximum(last - first, 0)
    total = int(count.sum())
    t_idx = np.repeat(np.arange(tl.size, dtype=IntDType), count)
    run_start = np.cumsum(count) - count
    s_idx = (np.arange(total, dtype=IntDType) - np.repeat(run_start, count) + np.repeat(first, count)).astype(IntDType)
    length = np.maximum(0.0, np.minimum(su[s_idx], tu[t_idx]) - np.maximum(sl[s_idx], tl[t_idx]))
    keep = length > 0.0
    return s_idx[keep], t_idx[keep], length[keep]


def _by_target_then_source(source_index, target_index, weights):
    """Rows of the weight matrix are target cells; within a row we order by source index (the reference's
    own within-row order is whatever its non-stable argsort leaves, structured.py:333)."""
    order = np.lexsort((weights, source_index, target_index))
    return source_index[order], target_index[order], weights[order]


def _axis_csr(source_index, target_index, weights, n_target):
    """(indptr, source, weight) of one axis; entries already ordered by (target, source)."""
    indptr = np.zeros(n_target + 1, dtype=np.int64)
    np.cumsum(np.bincount(target_index, minlength=n_target), out=indptr[1:])
    return indptr, np.ascontiguousarray(source_index, dtype=np.int64), np.ascontiguousarray(weights, dtype=np.float64)


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

    def flip_if_needed(self, index):
        return self.size - index - 1 if self.flipped else index

    # ---- structured -> structured, one axis (all return triplets ordered by target, then source)
    def overlap(self, other: "StructuredGrid1d", relative: bool):
        """Overlap length of every (source, target) cell pair (structured.py:183-206, :335-356)."""
        source_index, target_index, weights = _overlap_1d(self.bounds, other.bounds)
        source_index = self.flip_if_needed(source_index)
        target_index = other.flip_if_needed(target_index)
        if relative:
            # as the reference: ``length`` is in ascending-coordinate order, the index already flipped (:354-355)
            weights = weights / self.length[source_index]
        return _by_target_then_source(source_index, target_index, weights)

    def valid_nodes_within_bounds(self, other: "StructuredGrid1d"):
        """Source cell containing each target midpoint (structured.py:118-156)."""
        start = np.searchsorted(self.bounds[:, 0], other.midpoints, side=self.side)
        end = np.searchsorted(self.bounds[:, 1], other.midpoints, side=self.side)
        valid = (start == (end + 1)) & (other.midpoints > self.bounds[0, 0]) & (other.midpoints < self.bounds[-1, 1])
        valid_other_index = np.arange(other.size, dtype=IntDType)[valid]
        valid_self_index = end[valid].astype(IntDType)
        return self.flip_if_needed(valid_self_index), other.flip_if_needed(valid_other_index)

    def locate_centroids(self, other: "StructuredGrid1d"):
        """structured.py:358-377."""
        source_index, target_index = self.valid_nodes_within_bounds(other)
        weights = np.ones(source_index.size, dtype=float)
        return _by_target_then_source(source_index, target_index, weights)

    def linear_weights(self, other: "StructuredGrid1d"):
        """
        Two-point linear interpolation between source midpoints (structured.py:379-403, :201-315): each
        located target gets (source, w) and (neighbour, 1 - w); outside the outermost midpoints the
        neighbour collapses onto the source cell itself (weights 0 and 1).
        """
        if self.midpoints.size < 2:
            raise ValueError(
                f"Coordinate {self.name} has size: {self.midpoints.size}. "
                "At least two points are required for interpolation."
            )
        source_index, target_index = self.valid_nodes_within_bounds(other)
        src_mid = self.flip_if_needed(source_index)  # position in the ascending midpoints
        tgt_mid = other.flip_if_needed(target_index)
        step = np.where(other.midpoints[tgt_mid] <= self.midpoints[src_mid], -1, 1)
        neighbour_mid = np.clip(src_mid + step, 0, self.midpoints.size - 1)
        step = neighbour_mid - src_mid
        length = other.midpoints[tgt_mid] - self.midpoints[src_mid]
        total_length = self.midpoints[neighbour_mid] - self.midpoints[src_mid]
        total_length[total_length == 0] = 1
        weights = 1 - (length / total_length)
        weights[step == 0] = 0.0
        if self.flipped:
            step = -step
        pair_source = np.column_stack((source_index, source_index + step)).ravel()
        pair_target = np.repeat(target_index, 2)
        pair_weights = np.column_stack((weights, 1.0 - weights)).ravel()
        valid = (pair_source <= self.size - 1) & (pair_source >= 0)
        return _by_target_then_source(pair_source[valid], pair_target[valid], pair_weights[valid])


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
                # (the quads are generated on the device; host copies only if somebody asks for them)
                ugrid2d = Ugrid2d.from_structured_bounds_device(
                    self.xbounds.directional_bounds, self.ybounds.directional_bounds
                )
                self._unstructured = UnstructuredGrid2d(ugrid2d)
            return self._unstructured
        raise TypeError(f"Cannot convert StructuredGrid2d to {matched_type.__name__}")

    # ---- structured -> structured (structured.py:503-601)
    def _axes(self, other: "StructuredGrid2d", kind: str, relative: bool = False):
        if kind == "overlap":
            return self.ybounds.overlap(other.ybounds, relative), self.xbounds.overlap(other.xbounds, relative)
        if kind == "locate_centroids":
            return self.ybounds.locate_centroids(other.ybounds), self.xbounds.locate_centroids(other.xbounds)
        if kind == "linear_weights":
            return self.ybounds.linear_weights(other.ybounds), self.xbounds.linear_weights(other.xbounds)
        raise ValueError(kind)

    def _outer_host(self, other, axes):
        """broadcast_sorted (structured.py:503-531) on the host: (source_index, target_index, weights) ordered
        by target cell, then source cell.  O(P) numpy; the regridders use ``_outer_device`` instead."""
        (sy, ty, wy), (sx, tx, wx) = axes
        iy, ix = np.meshgrid(np.arange(sy.size), np.arange(sx.size), indexing="ij")
        iy, ix = iy.ravel(), ix.ravel()
        source_index = (sy[iy] * self.xbounds.size + sx[ix]).astype(IntDType)
        target_index = (ty[iy] * other.xbounds.size + tx[ix]).astype(IntDType)
        weights = wy[iy] * wx[ix]
        return _by_target_then_source(source_index, target_index, weights)

    def _outer_device(self, other, axes) -> "engine.DeviceOuter":
        """The weights in factored form on the device: applied matrix-free, product CSR only on request."""
        (sy, ty, wy), (sx, tx, wx) = axes
        return engine.DeviceOuter(
            _axis_csr(sy, ty, wy, other.ybounds.size), self.ybounds.size,
            _axis_csr(sx, tx, wx, other.xbounds.size), self.xbounds.size,
        )

    def overlap(self, other: "StructuredGrid2d", relative: bool):
        return self._outer_host(other, self._axes(other, "overlap", relative))

    def overlap_device(self, other: "StructuredGrid2d", relative: bool):
        """The (relative) overlap weights as a device CSR: rows = target cells (row-major y, x)."""
        return self._outer_device(other, self._axes(other, "overlap", relative))

    def locate_centroids(self, other: "StructuredGrid2d", tolerance=None):
        return self._outer_host(other, self._axes(other, "locate_centroids"))

    def locate_centroids_device(self, other: "StructuredGrid2d", tolerance=None):
        return self._outer_device(other, self._axes(other, "locate_centroids"))

    def linear_weights(self, other: "StructuredGrid2d"):
        return self._outer_host(other, self._axes(other, "linear_weights"))

    def linear_weights_device(self, other: "StructuredGrid2d"):
        return self._outer_device(other, self._axes(other, "linear_weights"))

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
