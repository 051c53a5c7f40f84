"""
Reduction methods of the regridders: the names of xugrid/regrid/reduce.py:254-272 mapped onto
the reducer ids of the HIP apply kernels (include/xugrid_amd.h, xugrid_amd/csrc/xr_apply.hip).

The reference hands Python functions ``f(values, weights, workspace)`` to numba; here a method
is a small descriptor ``Method(name, method_id, percentile)`` that selects a kernel
specialisation.  Arbitrary Python callables (a numba feature, examples/overlap_regridder.py:105-169)
cannot run on the device: they are the caller's own host code and run on the host over the engine's
weights, in the loop of make_regrid (regrid/regridder.py: _make_host_regrid).  No BUILT-IN reducer has a
host path -- there is no CPU fallback.
"""
from typing import NamedTuple

from .engine import METHOD_IDS


class Method(NamedTuple):
    name: str
    method_id: int
    percentile: float = 0.0


def create_percentile_method(p: float) -> Method:
    """reduce.py:241-251 -- raises ValueError outside [0, 100]."""
    if not (0.0 <= p <= 100.0):
        raise ValueError(f"percentile must be in the range [0, 100], received: {p}")
    return Method(f"p{p:g}", METHOD_IDS["percentile"], float(p))


def _simple(name):
    return Method(name, METHOD_IDS[name])


ABSOLUTE_OVERLAP_METHODS = {
    "mean": _simple("mean"),
    "harmonic_mean": _simple("harmonic_mean"),
    "geometric_mean": _simple("geometric_mean"),
    "sum": _simple("sum"),
    "minimum": _simple("minimum"),
    "maximum": _simple("maximum"),
    "mode": _simple("mode"),
    "median": Method("median", METHOD_IDS["percentile"], 50.0),
    "max_overlap": _simple("max_overlap"),
}
for _p in (5, 10, 25, 50, 75, 90, 95):
    ABSOLUTE_OVERLAP_METHODS[f"p{_p}"] = Method(f"p{_p}", METHOD_IDS["percentile"], float(_p))

RELATIVE_OVERLAP_METHODS = {
    "conductance": Method("conductance", METHOD_IDS["conductance"]),
    "first_order_conservative": Method("first_order_conservative", METHOD_IDS["first_order_conservative"]),
}
