"""
Host-side sparse containers with the field names of the reference
(xugrid/core/sparse.py:22-137): ``MatrixCOO`` and ``MatrixCSR`` NamedTuples of numpy arrays.
The device-resident counterpart is ``xugrid_amd.engine.DeviceCSR``.
"""
from typing import NamedTuple

import numpy as np

IntDType = np.intp


class MatrixCOO(NamedTuple):
    """Triplet matrix: ``row`` = target index, ``col`` = source index (regridder.py:290-295)."""

    data: np.ndarray
    row: np.ndarray
    col: np.ndarray
    n: int
    m: int
    nnz: int

    @staticmethod
    def from_triplet(row, col, data, n=None, m=None) -> "MatrixCOO":
        if n is None:
            n = row.max() + 1
        if m is None:
            m = col.max() + 1
        return MatrixCOO(data, row, col, n, m, data.size)

    def to_csr(self) -> "MatrixCSR":
        """Rows must already be sorted (core/sparse.py:65); only indptr is computed."""
        counts = np.bincount(self.row, minlength=self.n)
        indptr = np.zeros(counts.size + 1, dtype=IntDType)
        np.cumsum(counts, out=indptr[1:])
        return MatrixCSR(self.data, self.col, indptr, self.n, self.m, self.nnz)


class MatrixCSR(NamedTuple):
    """Compressed sparse rows: entries of target ``i`` are ``indptr[i]:indptr[i + 1]``."""

    data: np.ndarray
    indices: np.ndarray
    indptr: np.ndarray
    n: int
    m: int
    nnz: int

    @staticmethod
    def from_triplet(row, col, data, n=None, m=None) -> "MatrixCSR":
        return MatrixCOO.from_triplet(row, col, data, n, m).to_csr()

    def to_coo(self) -> MatrixCOO:
        row = np.repeat(np.arange(self.n, dtype=IntDType), np.diff(self.indptr))
        return MatrixCOO(self.data, row, self.indices, self.n, self.m, self.nnz)


# Row helpers of the reference (core/sparse.py:129-158; numba-inlined there, plain Python here: they serve host-side
# code that walks a downloaded matrix -- the device kernels index ``indptr`` directly).
def nzrange(A: MatrixCSR, row: int) -> range:
    """The positions of the entries of one row."""
    return range(int(A.indptr[row]), int(A.indptr[row + 1]))


def row_slice(A: MatrixCSR, row: int) -> slice:
    """The slice of ``indices`` / ``data`` holding one row."""
    return slice(int(A.indptr[row]), int(A.indptr[row + 1]))


def columns_and_values(A: MatrixCSR, slice):  # noqa: A002  (the reference's argument name)
    return zip(A.indices[slice], A.data[slice])
