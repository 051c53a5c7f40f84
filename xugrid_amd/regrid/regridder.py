"""
The regridder classes -- same names, constructor signatures, defaults, methods and error
behaviour as xugrid/regrid/regridder.py (classes :99-659), with the two hot loops replaced by
the HIP engine:

  * weight construction  (``_compute_weights`` :386-398, :428-436, :624-638)  ->  device kernels
    behind ``UnstructuredGrid2d.overlap_device / locate_centroids / barycentric``
  * apply                (``make_regrid(f)._regrid`` :41-67, COO ``_regrid`` :400-409)  ->
    ``DeviceCSR.apply`` / ``engine.apply_coo``

Weights live in HBM (``self._device_weights``) and are only downloaded when somebody asks for them
(``weights``, ``to_dataset``, ``weights_as_dataframe``).  Arrays in, arrays out: ``regrid`` takes
a numpy array whose trailing axes are the source grid's (``(..., n_face)`` or ``(..., ny, nx)``,
regridder.py:143-195) and returns float64 with the target's trailing shape.  xarray wrapping is
left to the caller (xarray is optional and absent in this image).
"""
import abc
from typing import Optional, Union

import numpy as np

from .. import engine
from ..reduce import ABSOLUTE_OVERLAP_METHODS, RELATIVE_OVERLAP_METHODS, Method, create_percentile_method
from ..sparse import MatrixCOO, MatrixCSR
from ..ugrid2d import DeviceUgrid2d, Ugrid2d
from . import persist
from .structured import Raster, StructuredGrid2d
from .unstructured import UnstructuredGrid2d


def setup_grid(obj, **kwargs):
    """regridder.py:72-80."""
    if isinstance(obj, (UnstructuredGrid2d, StructuredGrid2d)):
        return obj
    if isinstance(obj, Ugrid2d) or (hasattr(obj, "grid") and isinstance(getattr(obj, "grid"), Ugrid2d)):
        return UnstructuredGrid2d(obj)
    if isinstance(obj, Raster):
        return StructuredGrid2d(obj, name_y=kwargs.get("name_y", "y"), name_x=kwargs.get("name_x", "x"))
    if _is_xarray(obj):
        return StructuredGrid2d(obj, name_y=kwargs.get("name_y", "y"), name_x=kwargs.get("name_x", "x"))
    raise TypeError(
        f"Expected Ugrid2d, UgridDataArray-like or Raster/xarray.DataArray, received: {type(obj).__name__}"
    )


def _is_xarray(obj):
    mod = type(obj).__module__ or ""
    return mod.startswith("xarray") and hasattr(obj, "coords")


def convert_to_match(source, target):
    """regridder.py:83-96: a structured pair stays structured (separable weights); any pair with an
    unstructured member is promoted to UnstructuredGrid2d (rasters become quads)."""
    if isinstance(source, StructuredGrid2d) and isinstance(target, StructuredGrid2d):
        return source, target
    return source.convert_to(UnstructuredGrid2d), target.convert_to(UnstructuredGrid2d)


def _make_host_regrid(func):
    """make_regrid (regridder.py:34-69) for a caller-supplied reduction ``f(values, weights, workspace) -> float``: the same
    loop -- output NaN-initialised, the row's source values copied into the first workspace row in the order of the row's
    indices, the second workspace row handed over uninitialised, empty rows skipped -- run by the interpreter (the reference
    compiles the callable with numba, which is used here too when it is importable).  O(nnz) Python calls: meant for what
    the built-in reducers do not cover, not for million-cell grids."""
    try:
        import numba  # noqa: F401  (absent from this image; the reference's own path when present)

        f = numba.njit(func)
    except Exception:  # noqa: BLE001
        f = func

    def _regrid(source, A, size):
        n_extra = source.shape[0]
        out = np.full((n_extra, size), np.nan)
        indptr = np.asarray(A.indptr)
        n_work = int(np.diff(indptr).max()) if A.n > 0 else 0
        workspace = np.empty((2, max(n_work, 1)), dtype=np.float64)
        indices, data = np.asarray(A.indices), np.asarray(A.data, dtype=np.float64)
        rows = np.flatnonzero(indptr[1:] > indptr[:-1])
        for extra_index in range(n_extra):
            source_flat = source[extra_index]
            for target_index in rows:
                s, e = int(indptr[target_index]), int(indptr[target_index + 1])
                values = workspace[0, : e - s]
                values[:] = source_flat[indices[s:e]]
                out[extra_index, target_index] = f(values, data[s:e], workspace[1, : e - s])
        return out

    return _regrid


class BaseRegridder(abc.ABC):
    _METHODS = {}

    def __init__(self, source, target, tolerance: Optional[float] = None):
        self._source = setup_grid(source)
        self._target = setup_grid(target)
        self._weights = None
        self._device_weights = None
        self._compute_weights(self._source, self._target, tolerance)

    @abc.abstractmethod
    def _compute_weights(self, source, target, tolerance: Optional[float] = None):
        pass

    # ---- method selection (regridder.py:124-141)
    def _setup_regrid(self, func) -> None:
        if isinstance(func, str):
            try:
                self._method = self._METHODS[func]
            except KeyError as e:
                raise ValueError(
                    "Invalid regridding method. Available methods are: {}".format(self._METHODS.keys())
                ) from e
        elif isinstance(func, Method):
            self._method = func
        elif callable(func):
            # the caller's OWN reduction (regridder.py:136-137; examples/overlap_regridder.py:105-169): Python code cannot run on
            # the device.  The weights stay the engine's; the row loop of make_regrid (regridder.py:41-67) hands the callable
            # (values, weights, workspace) per non-empty target row on the host -- not a fallback of any built-in reducer.
            self._method = None
            self._custom = _make_host_regrid(func)
        else:
            raise TypeError(f"method must be string or callable, received: {type(func).__name__}")

    # ---- apply
    def _regrid(self, source: np.ndarray, size: int) -> np.ndarray:
        if self._method is None:
            w = self._ensure_host_weights()
            if isinstance(w, MatrixCOO):
                # (converted once per weight matrix, not per call)
                cached = getattr(self, "_custom_csr", None)
                if cached is None or cached[0] is not w:
                    cached = self._custom_csr = (w, w.to_csr())
                w = cached[1]
            if w.n != size:
                raise ValueError(f"the weights have {w.n} rows, the target grid {size} cells")
            return self._custom(np.asarray(source, dtype=np.float64), w, size)
        out = self._ensure_device_weights().apply(source, self._method.method_id, self._method.percentile)
        if out.shape[1] != size:
            raise ValueError(f"the weights have {out.shape[1]} rows, the target grid {size} cells")
        return out

    def _regrid_array(self, source):
        """regridder.py:143-195 on plain ndarrays."""
        if not isinstance(source, np.ndarray):
            raise TypeError(f"Expected numpy.ndarray. Received: {type(source).__name__}")
        source_grid = self._source
        if source.ndim < source_grid.ndim or source.shape[source.ndim - source_grid.ndim:] != source_grid.shape:
            raise ValueError(
                f"data does not contain regridder source dimensions: trailing shape {source_grid.shape} expected, "
                f"received {source.shape}"
            )
        first_dims_shape = source.shape[: source.ndim - source_grid.ndim]
        if source.ndim == source_grid.ndim:
            source = source[np.newaxis]
        source = source.reshape((-1, source_grid.size))
        size = self._target.size
        out = self._regrid(source, size)
        return out.reshape(first_dims_shape + self._target.shape)

    def _regrid_device(self, data, info):
        """``_regrid_array`` for data that already lives in HBM (a torch tensor on the GPU, anything with
        ``__cuda_array_interface__``): same layout contract (regridder.py:143-195 -- the spatial dims last, the leading ones
        flattened to K), device pointers straight into the apply kernels, and a result of the same kind on the device: nothing
        crosses PCIe.  The first call of a regridder whose weights are still deferred (device grids, see ``_compute_overlap``)
        builds them in the same engine call (xr_overlap_apply_dev)."""
        ptr, shape, dtype = info
        if getattr(self, "_method", True) is None:
            raise TypeError("a custom Python reduction runs on the host: pass host arrays")
        if dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise TypeError(f"device data must be float64 or float32, received {dtype}")
        source_grid = self._source
        nd = source_grid.ndim
        if len(shape) < nd or tuple(shape[len(shape) - nd:]) != tuple(source_grid.shape):
            raise ValueError(
                f"data does not contain regridder source dimensions: trailing shape {source_grid.shape} expected, "
                f"received {tuple(shape)}"
            )
        first_dims_shape = tuple(shape[: len(shape) - nd])
        K = int(np.prod(first_dims_shape, dtype=np.int64)) if first_dims_shape else 1
        out, out_ptr = engine.empty_like_device(data, first_dims_shape + tuple(self._target.shape), np.float64)
        engine.sync_producer(data)
        dtype_id = engine.XR_F64 if dtype == np.dtype(np.float64) else engine.XR_F32
        method_id, percentile = self._apply_method()
        deferred = getattr(self, "_deferred", None)
        if self._device_weights is None and deferred is not None:
            source, target, relative = deferred
            self._device_weights = source.ugrid_topology.device_mesh.overlap_apply_dev(
                target.ugrid_topology.device_mesh, ptr, dtype_id, K, out_ptr, method_id, percentile, relative=relative)
            self._deferred = None
        else:
            weights = self._ensure_device_weights()
            rows = getattr(weights, "n", self._target.size)
            if rows != self._target.size:
                raise ValueError(f"the weights have {rows} rows, the target grid {self._target.size} cells")
            weights.apply_dev(ptr, dtype_id, K, out_ptr, method_id, percentile)
        engine.dev_sync()  # (the result is complete when the call returns, whatever stream its consumer uses)
        return out

    def _apply_method(self):
        return self._method.method_id, self._method.percentile

    def regrid(self, data):
        """
        Regrid ``data`` from the source topology to the target topology; additional leading
        dimensions (time, layer ...) are regridded in one batched device call.

        data: np.ndarray ``(..., n_face)`` / ``(..., ny, nx)``, an object exposing ``.values``
        (e.g. an xarray.DataArray, whose dims must end with the source dims), or an array that already lives on the
        device (torch tensor on the GPU / ``__cuda_array_interface__``): the result then stays there too.
        """
        if isinstance(data, np.ndarray):
            return self._regrid_array(data)
        info = engine.device_array_info(data)
        if info is not None:
            return self._regrid_device(data, info)
        if hasattr(data, "values") and hasattr(data, "dims"):
            values = np.asarray(data.values)
            dims = tuple(data.dims)
            # regridder.py:231-251: the source dims come from the DATA, not from the regridder's source grid -- after
            # from_weights / from_dataset that grid is called "__source" and its face dim "__source_nFaces" (the
            # reference's FIXME at :231-236).  Structured data: ("y", "x") (:239); unstructured data: the core
            # dimension of the data's own grid (:242).  The dims are looked up BY NAME and moved to the end
            # (apply_ufunc's input_core_dims, :197-210); a bare trailing-shape check would silently regrid the wrong
            # axes of e.g. (x, y) data with nx == ny.
            source_dims = self._data_source_dims(data, dims)
            missing = set(source_dims) - set(dims)
            if missing:
                raise ValueError(f"data does not contain regridder source dimensions: {missing}")
            other = [d for d in dims if d not in source_dims]
            values = np.transpose(values, [dims.index(d) for d in other] + [dims.index(d) for d in source_dims])
            return self._regrid_array(np.ascontiguousarray(values))
        raise TypeError(f"Expected DataArray or UgridDataAray, received: {type(data).__name__}")

    def _data_source_dims(self, data, dims):
        if isinstance(self._source, StructuredGrid2d):
            named = tuple(self._source.dims)
            return named if set(named) <= set(dims) else ("y", "x")
        grid = getattr(getattr(data, "ugrid", None), "grid", None) or getattr(data, "grid", None)
        core = getattr(grid, "core_dimension", None) or getattr(grid, "face_dimension", None)
        if isinstance(core, str):
            return (core,)
        # a bare DataArray-like over an unstructured source: the regridder's own face dim when the data names it,
        # else the data's last dim (the layout contract of _regrid_array, regridder.py:145-163)
        named = tuple(self._source.dims)
        if set(named) <= set(dims):
            return named
        return (dims[-1],) if dims else ()

    # ---- weights access / persistence (regridder.py:264-361)
    def _build_deferred(self):
        deferred = getattr(self, "_deferred", None)
        if self._device_weights is None and deferred is not None:
            source, target, relative = deferred
            self._device_weights = source.overlap_device(target, relative=relative)
            self._deferred = None

    def _ensure_host_weights(self):
        if self._weights is None:
            self._build_deferred()
            if self._device_weights is None:
                raise ValueError("Weights have not been computed yet.")
            data, indices, indptr = self._device_weights.download()
            dw = self._device_weights
            self._weights = MatrixCSR(data, indices, indptr, dw.n, dw.m, dw.nnz)
        return self._weights

    def _ensure_device_weights(self):
        self._build_deferred()
        if self._device_weights is None:
            w = self._weights
            if w is None:
                raise ValueError("Weights have not been computed yet.")
            if isinstance(w, MatrixCOO):
                w = w.to_csr()
            self._device_weights = engine.DeviceCSR.from_arrays(w.data, w.indices, w.indptr, w.n, w.m)
            self._attach_row_keys(self._device_weights)
        return self._device_weights

    def _attach_row_keys(self, device_weights):
        """Uploaded (cached) weights carry no geometry: hand the engine a coarse Morton key per target cell so
        that applies of many variables can regroup the rows into compact tiles (pure locality hint)."""
        target = self._target
        try:
            if isinstance(target, StructuredGrid2d):
                yy, xx = np.meshgrid(target.ybounds.index, target.xbounds.index, indexing="ij")
                centres = np.column_stack([xx.ravel(), yy.ravel()])
            else:
                centres = target.ugrid_topology.centroids
        except Exception:
            return
        if centres.shape[0] == device_weights.n and device_weights.n >= 4096:
            keys, key_range = engine.morton_row_keys(centres)
            device_weights.set_row_keys(keys, key_range)

    def to_dataset(self) -> dict:
        """Weights + source + target topology as a flat dict of arrays, with the variable names of
        regridder.py:264-271 (``__regrid_data/indices/indptr/n/m/nnz`` or ``row/col`` for COO)."""
        w = self._ensure_host_weights()
        ds = {f"__regrid_{k}": v for k, v in zip(w._fields, w)}
        ds.update(self._source.to_dataset("__source"))
        ds.update(self._target.to_dataset("__target"))
        return ds

    @property
    def weights(self):
        return self.to_dataset()

    def weights_as_dataframe(self):
        """Three columns: target_index, source_index, weight (regridder.py:273-296)."""
        import pandas as pd

        matrix = self._ensure_host_weights()
        if isinstance(matrix, MatrixCSR):
            matrix = matrix.to_coo()
        return pd.DataFrame({"target_index": matrix.row, "source_index": matrix.col, "weight": matrix.data})

    @staticmethod
    def _csr_from_dataset(dataset) -> MatrixCSR:
        return MatrixCSR(
            np.asarray(dataset["__regrid_data"]),
            np.asarray(dataset["__regrid_indices"]),
            np.asarray(dataset["__regrid_indptr"]),
            int(np.asarray(dataset["__regrid_n"]).item()),
            int(np.asarray(dataset["__regrid_m"]).item()),
            int(np.asarray(dataset["__regrid_nnz"]).item()),
        )

    @staticmethod
    def _coo_from_dataset(dataset) -> MatrixCOO:
        return MatrixCOO(
            np.asarray(dataset["__regrid_data"]),
            np.asarray(dataset["__regrid_row"]),
            np.asarray(dataset["__regrid_col"]),
            int(np.asarray(dataset["__regrid_n"]).item()),
            int(np.asarray(dataset["__regrid_m"]).item()),
            int(np.asarray(dataset["__regrid_nnz"]).item()),
        )

    @classmethod
    @abc.abstractmethod
    def _weights_from_dataset(cls, dataset):
        """Return either COO or CSR weights."""

    @classmethod
    def _weights_from_reference(cls, dataset):
        """The same from the reference's layout (CSR unless the class stores COO)."""
        return persist.csr_from_reference(dataset)

    @staticmethod
    def _grid_from_dataset(dataset, name):
        kind = dataset[name + "_type"]
        kind = kind if isinstance(kind, str) else str(np.asarray(kind).item())
        if kind == "UnstructuredGrid2d":
            return setup_grid(Ugrid2d.from_dataset(dataset, name))
        return StructuredGrid2d.from_dataset(dataset, name)

    # ---- the reference's own dataset layout (regridder.py:264-271, :334-361; regrid/persist.py)
    def to_reference_dataset(self) -> "persist.RefDataset":
        """``to_dataset()`` in the layout xugrid itself writes: attrs-typed ``__source_type`` / ``__target_type``
        markers, UGRID topology variables, ascending midpoints + bounds for rasters -- a ``RefDataset`` of
        (dims, data, attrs) variables (``.to_xarray()`` when xarray is present; ``.save(path)`` otherwise).
        xugrid's ``Regridder.from_dataset`` / ``from_weights`` read it as it is."""
        ds = persist.weights_to_reference(self._ensure_host_weights())
        ds.merge(persist.grid_to_reference(self._source, "__source"))
        ds.merge(persist.grid_to_reference(self._target, "__target"))
        return ds

    @classmethod
    def from_reference_dataset(cls, dataset, target=None, **kwargs):
        """Reconstruct from a dataset in the reference's layout: a ``RefDataset`` or a real ``xr.Dataset`` written by
        xugrid (``regridder.to_dataset()``).  ``target``: as in ``from_weights`` (regridder.py:334-347); without it
        the target is read from the dataset too (``from_dataset``, :350-361).  kwargs: ``method`` where the class
        takes one."""
        if target is None:
            if persist.grid_kind(dataset, "__target") != "UnstructuredGrid2d":
                # (the reference leaves ``target`` unbound for structured targets, :355-360; here they are read)
                target = persist.structured2d_from_reference(dataset, "__target")
            else:
                target = persist.ugrid2d_from_reference(dataset, "__target")
        return cls.from_weights(dataset, target, **kwargs)

    @classmethod
    def from_weights(cls, weights, target):
        instance = cls.__new__(cls)
        instance._device_weights = None
        instance._target = setup_grid(target)
        if persist.is_reference_layout(weights):
            instance._weights = cls._weights_from_reference(weights)
            instance._source = persist.grid_from_reference(weights, "__source")
        else:
            instance._weights = cls._weights_from_dataset(weights)
            instance._source = cls._grid_from_dataset(weights, "__source")
        w = instance._weights
        if w.n != instance._target.size:
            raise ValueError(f"the weights have {w.n} rows, the target grid {instance._target.size} cells")
        if w.m != instance._source.size:
            raise ValueError(f"the weights have {w.m} columns, the source grid {instance._source.size} cells")
        return instance

    # ---- file persistence (xarray / netCDF are optional and absent here).  layout="flat": the flat dict of
    # ``to_dataset`` (the reference's weight variable names, this package's own grid variables); layout="reference":
    # ``to_reference_dataset()`` -- the reference's variables, dims and attrs one to one, i.e. what
    # ``xr.Dataset.to_netcdf`` would carry -- in an .npz with a JSON sidecar for dims / attrs.
    def to_file(self, path, layout: str = "flat") -> None:
        """Write the weights and both grids to a NumPy ``.npz`` archive."""
        if layout == "reference":
            self.to_reference_dataset().save(path)
            return
        if layout != "flat":
            raise ValueError(f'layout must be "flat" or "reference", received {layout!r}')
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in self.to_dataset().items()})

    @classmethod
    def from_file(cls, path):
        """Reconstruct the regridder (cached weights + both grids) from ``to_file`` output (either layout)."""
        with np.load(path, allow_pickle=False) as archive:
            if "__meta__" in archive.files:
                return cls.from_reference_dataset(persist.RefDataset.load(path))
            dataset = {k: (archive[k].item() if archive[k].ndim == 0 and archive[k].dtype.kind in "US" else archive[k])
                       for k in archive.files}
        return cls.from_dataset(dataset)

    @classmethod
    def from_dataset(cls, dataset):
        """Reconstruct the regridder from ``to_dataset()`` output (regridder.py:350-361); a dataset in the
        reference's own layout (``to_reference_dataset``, or an ``xr.Dataset`` written by xugrid) is recognised."""
        if persist.is_reference_layout(dataset):
            return cls.from_reference_dataset(dataset)
        target = cls._grid_from_dataset(dataset, "__target")
        return cls.from_weights(dataset, target)


class CentroidLocatorRegridder(BaseRegridder):
    """
    Regrids by locating the centroids of the target faces in the source grid
    (regridder.py:364-422).  If a centroid lies exactly on an edge between two faces the face with
    the lowest index is used (the reference leaves the choice unspecified, :369-370).

    The weights are the reference's MatrixCOO (one (target, source, 1.0) triplet per located centroid); on
    the device they live as CSR rows with at most one entry and are applied with the ``select`` kernel
    (= the COO scatter ``out[k, row] = source[k, col]``, NaN included).
    """

    def _compute_weights(self, source, target, tolerance: Optional[float] = None):
        source, target = convert_to_match(source, target)
        self._device_weights = source.locate_centroids_device(target, tolerance)
        self._weights = None

    def _apply_method(self):
        return engine.METHOD_IDS["select"], 0.0

    def _regrid(self, source, size):
        A = self._weights
        if self._device_weights is None and A is not None and A.row.size and (np.diff(A.row) < 0).any():
            # externally supplied, unsorted COO weights: plain scatter (regridder.py:400-409)
            return engine.apply_coo(A.row, A.col, size, source)
        out = self._ensure_device_weights().apply(source, engine.METHOD_IDS["select"], 0.0)
        if out.shape[1] != size:
            raise ValueError(f"the weights have {out.shape[1]} rows, the target grid {size} cells")
        return out

    def _ensure_host_weights(self):
        if self._weights is None:
            if self._device_weights is None:
                raise ValueError("Weights have not been computed yet.")
            dw = self._device_weights
            data, indices, indptr = dw.download()
            row = np.repeat(np.arange(dw.n, dtype=indices.dtype), np.diff(indptr))
            self._weights = MatrixCOO(data, row, indices, dw.n, dw.m, dw.nnz)
        return self._weights

    def _ensure_device_weights(self):
        if self._device_weights is None:
            w = self._weights
            if w is None:
                raise ValueError("Weights have not been computed yet.")
            self._device_weights = engine.DeviceCSR.from_triplet(w.row, w.col, w.data, w.n, w.m)
        return self._device_weights

    @classmethod
    def _weights_from_dataset(cls, dataset) -> MatrixCOO:
        return cls._coo_from_dataset(dataset)

    @classmethod
    def _weights_from_reference(cls, dataset) -> MatrixCOO:
        return persist.coo_from_reference(dataset)


class BaseOverlapRegridder(BaseRegridder, abc.ABC):
    def _compute_overlap(self, source, target, relative: bool) -> None:
        source, target = convert_to_match(source, target)
        self._weights = None
        if (isinstance(source, UnstructuredGrid2d) and isinstance(source.ugrid_topology, DeviceUgrid2d)
                and isinstance(target.ugrid_topology, DeviceUgrid2d)):
            # Both grids were made from device arrays: a pipeline that keeps its data in HBM.  The weights are built by the
            # first call that needs them -- ``regrid`` of device data does it in ONE engine call with the apply
            # (xr_overlap_apply_dev: the apply rides on the construction, the step bench.py times); ``weights``,
            # ``to_dataset`` or host data build them on their own.  (The reference builds in the constructor,
            # regridder.py:428-436; what is computed is the same.)
            self._device_weights = None
            self._deferred = (source, target, relative)
            return
        self._device_weights = source.overlap_device(target, relative=relative)

    @classmethod
    def _weights_from_dataset(cls, dataset) -> MatrixCSR:
        return cls._csr_from_dataset(dataset)


class OverlapRegridder(BaseOverlapRegridder):
    """
    Area-weighted regridding from the overlap of target and source faces
    (regridder.py:443-532).  Methods: ``mean, harmonic_mean, geometric_mean, sum, minimum,
    maximum, mode, median, max_overlap, p5, p10, p25, p50, p75, p90, p95`` and any
    ``OverlapRegridder.create_percentile_method(p)``.
    """

    _METHODS = ABSOLUTE_OVERLAP_METHODS

    def __init__(self, source, target, method: Union[str, Method] = "mean"):
        super().__init__(source=source, target=target)
        self._setup_regrid(method)

    def _compute_weights(self, source, target, tolerance: Optional[float] = None) -> None:
        self._compute_overlap(source, target, relative=False)

    @staticmethod
    def create_percentile_method(percentile: float) -> Method:
        return create_percentile_method(percentile)

    @classmethod
    def from_weights(cls, weights, target, method: Union[str, Method] = "mean"):
        instance = super().from_weights(weights, target)
        instance._setup_regrid(method)
        return instance


class RelativeOverlapRegridder(BaseOverlapRegridder):
    """
    As OverlapRegridder, but the overlap area is divided by the area of the source face
    (regridder.py:535-586); methods ``first_order_conservative`` (default) and ``conductance``.
    """

    _METHODS = RELATIVE_OVERLAP_METHODS

    def __init__(self, source, target, method: Union[str, Method] = "first_order_conservative"):
        super().__init__(source=source, target=target, tolerance=None)
        self._setup_regrid(method)

    def _compute_weights(self, source, target, tolerance: Optional[float] = None) -> None:
        self._compute_overlap(source, target, relative=True)

    @classmethod
    def from_weights(cls, weights, target, method: Union[str, Method] = "first_order_conservative"):
        instance = super().from_weights(weights, target)
        instance._setup_regrid(method)
        return instance


class BarycentricInterpolator(BaseRegridder):
    """
    Interpolates with barycentric weights in the centroidal Voronoi tessellation of the source
    grid, evaluated at the target face centroids (regridder.py:589-659).  The weights of a target
    sum to one, so the reducer is fixed to ``mean`` (NaN-aware renormalisation).
    """

    _METHODS = {"mean": ABSOLUTE_OVERLAP_METHODS["mean"]}

    def __init__(self, source, target, tolerance: Optional[float] = None, tree_order: bool = False):
        # Default = the reference's result (regridder.py:613-622, unstructured.py:175,193: weight slots paired with
        # the caller's vertex order).  tree_order (opt-in, not in the reference's signature): pair the slots of the
        # concave exterior Voronoi cells with the order the weights were computed in instead
        # (UnstructuredGrid2d.barycentric; DESIGN.md section 7: changes 0.1-0.7 % of the entries)
        self._tree_order = bool(tree_order)
        super().__init__(source, target, tolerance)
        self._setup_regrid("mean")

    def _compute_weights(self, source, target, tolerance: Optional[float] = None):
        source, target = convert_to_match(source, target)
        if isinstance(source, StructuredGrid2d):
            # regridder.py:628-630: linear interpolation between cell midpoints, per axis
            self._device_weights = source.linear_weights_device(target)
            self._weights = None
            return
        self._device_weights = source.barycentric_device(target, tolerance, tree_order=self._tree_order)
        self._weights = None

    @classmethod
    def from_weights(cls, weights, target):
        instance = super().from_weights(weights, target)
        instance._setup_regrid("mean")
        return instance

    @classmethod
    def _weights_from_dataset(cls, dataset) -> MatrixCSR:
        return cls._csr_from_dataset(dataset)
