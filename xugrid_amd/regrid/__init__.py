from .regridder import (  # noqa: F401
    BarycentricInterpolator,
    CentroidLocatorRegridder,
    OverlapRegridder,
    RelativeOverlapRegridder,
)
from .structured import Raster, StructuredGrid2d  # noqa: F401
from .unstructured import UnstructuredGrid2d  # noqa: F401
