"""
Golden vectors for the separable structured -> structured weight construction:
StructuredGrid2d.overlap / locate_centroids / linear_weights of xugrid 0.15.3
(xugrid/regrid/structured.py:24-601), executed from /root/reference.

Runs ONLY in the build container:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python -B /root/repo/tests/golden/gen_structured.py

structured.py itself imports xarray (absent here), so the two classes are lifted out of the file with
``ast`` at run time and executed against a tiny stand-in for the coordinate container they read
(``obj.indexes[name]`` = pandas Index, ``obj.coords``, ``obj[name].to_numpy()``); overlap_1d and
utils.broadcast are imported normally.  Only DATA is written (g9_structured.npz): for each case the raster
coordinates and the reference's triplets, canonically ordered by (target, source) because the reference's
own within-row order comes out of a non-stable argsort (structured.py:333, :527).
"""
import ast
import os
import sys
import types
import typing
import warnings

import numpy as np
import pandas as pd

warnings.simplefilter("ignore")
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

pkg = types.ModuleType("xugrid")
pkg.__path__ = [f"{REF}/xugrid"]
sys.modules["xugrid"] = pkg
from xugrid.regrid.overlap_1d import overlap_1d, overlap_1d_nd  # noqa: E402
from xugrid.regrid.utils import broadcast  # noqa: E402

tree = ast.parse(open(f"{REF}/xugrid/regrid/structured.py").read())
classes = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("StructuredGrid1d", "StructuredGrid2d")]
xr_stub = types.SimpleNamespace(DataArray=object, Dataset=object)
ns = {
    "np": np, "xr": xr_stub, "overlap_1d": overlap_1d, "overlap_1d_nd": overlap_1d_nd, "broadcast": broadcast,
    "Any": typing.Any, "Tuple": typing.Tuple, "Union": typing.Union, "FloatArray": np.ndarray, "IntArray": np.ndarray,
    "Ugrid2d": None, "UnstructuredGrid2d": None,
}
exec(compile(ast.Module(body=classes, type_ignores=[]), "structured_lifted", "exec"), ns)
StructuredGrid2d = ns["StructuredGrid2d"]


class _Var:
    def __init__(self, a):
        self.a = np.asarray(a)

    def to_numpy(self):
        return self.a


class FakeRaster:
    """What StructuredGrid1d.__init__ reads from a DataArray (structured.py:33-83)."""

    name = "fake"

    def __init__(self, x, y, dx=None, dy=None, xbounds=None, ybounds=None):
        self.indexes = {"x": pd.Index(np.asarray(x, dtype=float)), "y": pd.Index(np.asarray(y, dtype=float))}
        self.coords = {}
        for k, v in (("dx", dx), ("dy", dy), ("xbounds", xbounds), ("ybounds", ybounds)):
            if v is not None:
                self.coords[k] = np.asarray(v, dtype=float)

    def __getitem__(self, key):
        return _Var(self.coords[key])


def canon(s, t, w):
    order = np.lexsort((s, t))
    return s[order].astype(np.int64), t[order].astype(np.int64), w[order].astype(np.float64)


def mid(edges):
    e = np.asarray(edges, dtype=float)
    return 0.5 * (e[1:] + e[:-1])


def bounds(edges):
    e = np.asarray(edges, dtype=float)
    return np.column_stack((e[:-1], e[1:]))


def main():
    rng = np.random.default_rng(909)
    cases = {}

    def add(name, src, tgt):
        cases[name] = (src, tgt)

    # a: equidistant 3x3 (50 m) -> 4x4 shifted 25 m (tests/test_regrid/test_structured.py:108-204 style)
    add("a", dict(x=[50.0, 100.0, 150.0], y=[50.0, 100.0, 150.0]),
        dict(x=[25.0, 75.0, 125.0, 175.0], y=[25.0, 75.0, 125.0, 175.0]))
    # b: descending y on both (the usual north-up raster), finer target partially outside the source
    add("b", dict(x=np.arange(5.0, 100.0, 10.0), y=np.arange(95.0, 0.0, -10.0)),
        dict(x=np.arange(-7.5, 120.0, 5.0), y=np.arange(112.5, -10.0, -5.0)))
    # c: descending y source only, ascending target, coarser target
    add("c", dict(x=np.arange(0.5, 60.0, 1.0), y=np.arange(39.5, 0.0, -1.0)),
        dict(x=np.arange(3.5, 60.0, 7.0), y=np.arange(2.5, 40.0, 5.0)))
    # d: non-equidistant, explicit bounds (ascending: the reference takes explicit bounds as they are, so a
    # descending coordinate with bounds has no consistent meaning there and is not pinned)
    exs = np.cumsum(rng.uniform(0.5, 2.0, 41)) + 3.0
    eys = np.cumsum(rng.uniform(0.5, 2.0, 31)) - 7.0
    ext = np.cumsum(rng.uniform(0.8, 3.0, 26)) + 1.0
    eyt = np.cumsum(rng.uniform(0.3, 1.8, 36)) - 9.0
    add("d", dict(x=mid(exs), y=mid(eys), xbounds=bounds(exs), ybounds=bounds(eys)),
        dict(x=mid(ext), y=mid(eyt), xbounds=bounds(ext), ybounds=bounds(eyt)))
    # e: cell sizes given as dx (scalar, negative for the descending axis) / dy arrays
    dys = rng.uniform(0.5, 1.5, 24)
    ey = np.concatenate(([0.0], np.cumsum(dys)))
    add("e", dict(x=np.arange(0.25, 12.0, 0.5), y=mid(ey), dx=0.5, dy=dys),
        dict(x=np.arange(11.0, 0.0, -2.0), y=np.arange(1.0, 24.0, 2.0), dx=-2.0, dy=2.0))
    # f: identical grids
    add("f", dict(x=np.arange(0.5, 8.0), y=np.arange(6.5, 0.0, -1.0)), dict(x=np.arange(0.5, 8.0), y=np.arange(6.5, 0.0, -1.0)))
    # g: target entirely outside the source along x
    add("g", dict(x=np.arange(0.5, 8.0), y=np.arange(0.5, 5.0)), dict(x=np.arange(20.5, 25.0), y=np.arange(0.5, 5.0)))
    # h: aligned 2:1 refinement and coarsening (edges coincide exactly)
    add("h", dict(x=np.arange(1.0, 33.0, 2.0), y=np.arange(1.0, 21.0, 2.0)), dict(x=np.arange(0.5, 32.0, 1.0), y=np.arange(2.0, 20.0, 4.0)))

    # i: equidistant source descending on BOTH axes, non-equidistant ascending target given by dx/dy arrays
    dxi = rng.uniform(0.5, 2.5, 14)
    dyi = rng.uniform(0.5, 2.5, 11)
    exi = np.concatenate(([0.0], np.cumsum(dxi)))
    eyi = np.concatenate(([0.0], np.cumsum(dyi)))
    add("i", dict(x=np.arange(24.5, -2.0, -1.0), y=np.arange(19.25, -1.0, -1.5)),
        dict(x=mid(exi), y=mid(eyi), dx=dxi, dy=dyi))

    out = {"cases": np.array(sorted(cases))}
    for name, (src, tgt) in cases.items():
        s = StructuredGrid2d(FakeRaster(**src), "x", "y")
        t = StructuredGrid2d(FakeRaster(**tgt), "x", "y")
        for side, d in (("src", src), ("tgt", tgt)):
            for k, v in d.items():
                out[f"{name}_{side}_{k}"] = np.asarray(v, dtype=float)
        out[f"{name}_src_shape"] = np.array(s.shape)
        out[f"{name}_tgt_shape"] = np.array(t.shape)
        out[f"{name}_src_area"] = s.area
        for kind, fn in (
            ("overlap", lambda: s.overlap(t, relative=False)),
            ("relative", lambda: s.overlap(t, relative=True)),
            ("locate", lambda: s.locate_centroids(t, None)),
            ("linear", lambda: s.linear_weights(t)),
        ):
            si, ti, w = canon(*fn())
            out[f"{name}_{kind}_src"] = si
            out[f"{name}_{kind}_tgt"] = ti
            out[f"{name}_{kind}_w"] = w
            print(name, kind, si.size)
    np.savez_compressed(os.path.join(OUT, "g9_structured.npz"), **out)


if __name__ == "__main__":
    main()
