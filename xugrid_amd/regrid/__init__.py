from .regridder import (  # noqa: F401
    BarycentricInterpolator,
    CentroidLocatorRegridder,
    OverlapRegridder,
    RelativeOverlapRegridder,
)
from .gridder import NetworkGridder  # noqa: F401
from .network import Network1d  # noqa: F401
from .structured import Raster, StructuredGrid2d  # noqa: F401
from .unstructured import UnstructuredGrid2d  # noqa: F401
