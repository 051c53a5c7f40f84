"""
xugrid_amd -- MI355X-native engine for xugrid's regridding hot path (weight construction and the
sparse weight x data apply), behind the reference's own Regridder API.  See DESIGN.md.

The compute path is hand-written HIP for gfx950 in ``xugrid_amd/csrc`` exposed through the C ABI
of ``include/xugrid_amd.h``; this package is the thin Python host side.  There is no CPU
fallback: without ``libxugrid_amd.so`` and a HIP device every compute call raises.
"""
from . import engine, meshgen  # noqa: F401
from ._lib import XugridAmdError  # noqa: F401
from .celltree import CellTree2d  # noqa: F401
from .regrid import (  # noqa: F401
    BarycentricInterpolator,
    CentroidLocatorRegridder,
    NetworkGridder,
    OverlapRegridder,
    Raster,
    RelativeOverlapRegridder,
)
from .sparse import MatrixCOO, MatrixCSR  # noqa: F401
from .ugrid1d import Ugrid1d  # noqa: F401
from .ugrid2d import Ugrid2d  # noqa: F401

__version__ = "0.1.0"
