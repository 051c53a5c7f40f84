"""
``NetworkGridder``: data on the edges of a 1-D network onto the faces of a 2-D grid, weighted by the length of
every edge inside every face -- counterpart of xugrid/regrid/gridder.py:24-85.

The weights come from ONE device call (``xr_edge_length_csr``: spatial-index walk + Cyrus-Beck clip per
(edge, face) candidate + CSR assembly, ``csrc/xr_edges.hip``) that replaces ``celltree.intersect_edges``, the
norm of the intersections, the argsort and ``MatrixCSR.from_triplet`` (xugrid/regrid/unstructured.py:203-215,
gridder.py:66-73); they stay in HBM and are applied with the same reducer kernels as the overlap regridders.
"""
from typing import Union

import numpy as np

from ..reduce import ABSOLUTE_OVERLAP_METHODS, Method
from ..sparse import MatrixCSR
from ..ugrid1d import Ugrid1d
from .network import Network1d
from .regridder import BaseRegridder, setup_grid
from .unstructured import UnstructuredGrid2d


def convert_to_match(source, target):
    """gridder.py:14-21: the target (raster or mesh) is always handled as an unstructured grid."""
    return source, target.convert_to(UnstructuredGrid2d)


class NetworkGridder(BaseRegridder):
    """
    Network gridder for 2D grids.

    source: Ugrid1d (or a wrapper exposing ``.grid``); target: Ugrid2d, Raster or xarray.DataArray;
    method: one of ``mean, harmonic_mean, geometric_mean, sum, minimum, maximum, mode, median, max_overlap, p5 ...
    p95`` or a percentile ``Method`` (reduce.ABSOLUTE_OVERLAP_METHODS, gridder.py:35-37).
    """

    _METHODS = ABSOLUTE_OVERLAP_METHODS

    def __init__(self, source, target, method: Union[str, Method] = "mean"):
        self._source = Network1d(source)
        self._target = setup_grid(target)
        self._weights = None
        self._device_weights = None
        self._compute_weights(self._source, self._target, relative=False)
        self._setup_regrid(method)

    def _compute_weights(self, source, target, relative: bool = False) -> None:
        source, target = convert_to_match(source, target)
        if relative:  # (never requested by the reference either, gridder.py:49; the host path reproduces its formula)
            raise NotImplementedError("NetworkGridder uses absolute intersection lengths")
        self._device_weights = target.intersection_length_device(source)
        self._weights = None

    @property
    def weights(self):
        return self.to_dataset()

    @weights.setter
    def weights(self, weights: MatrixCSR):
        if not isinstance(weights, MatrixCSR):
            raise TypeError(f"Expected MatrixCSR, received: {type(weights).__name__}")
        self._weights = weights
        self._device_weights = None

    @classmethod
    def _weights_from_dataset(cls, dataset) -> MatrixCSR:
        return cls._csr_from_dataset(dataset)

    @staticmethod
    def _grid_from_dataset(dataset, name):
        kind = dataset[name + "_type"]
        kind = kind if isinstance(kind, str) else str(np.asarray(kind).item())
        if kind == "Network1d":
            return Network1d(Ugrid1d.from_dataset(dataset, name))
        return BaseRegridder._grid_from_dataset(dataset, name)

    @classmethod
    def from_weights(cls, weights, target, method: Union[str, Method] = "mean"):
        instance = super().from_weights(weights, target)
        instance._setup_regrid(method)
        return instance
