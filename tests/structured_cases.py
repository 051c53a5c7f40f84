"""Shared helpers of the structured (raster -> raster) tests: golden cases and random rasters."""
import numpy as np

KINDS = ("overlap", "relative", "locate", "linear")
AXIS_KEYS = ("x", "y", "dx", "dy", "xbounds", "ybounds")


def raster_kwargs(g, name, side):
    kw = {}
    for k in AXIS_KEYS:
        key = f"{name}_{side}_{k}"
        if key in g:
            v = g[key]
            kw[k] = v if v.ndim else float(v)
    return kw


def golden_triplets(g, name, kind):
    s, t, w = g[f"{name}_{kind}_src"], g[f"{name}_{kind}_tgt"], g[f"{name}_{kind}_w"]
    order = np.lexsort((w, s, t))
    return s[order], t[order], w[order]


def canon(s, t, w):
    order = np.lexsort((w, s, t))
    return np.asarray(s)[order], np.asarray(t)[order], np.asarray(w)[order]


def random_raster(rng, n_max=60, allow_flip=True):
    """kwargs of a random valid raster: per axis either equidistant (possibly descending) or
    non-equidistant ascending given by a size array or explicit bounds."""
    kw = {}
    for ax in ("x", "y"):
        n = int(rng.integers(2, n_max))
        mode = rng.integers(0, 4)
        x0 = rng.uniform(-5, 5)
        if mode == 0:  # equidistant, maybe descending, nothing else given
            d = rng.uniform(0.2, 2.0)
            mid = x0 + d * (np.arange(n) + 0.5)
            kw[ax] = mid[::-1].copy() if (allow_flip and rng.random() < 0.5) else mid
        elif mode == 1:  # equidistant with scalar d (negative when descending)
            d = rng.uniform(0.2, 2.0)
            mid = x0 + d * (np.arange(n) + 0.5)
            if allow_flip and rng.random() < 0.5:
                kw[ax], kw["d" + ax] = mid[::-1].copy(), -d
            else:
                kw[ax], kw["d" + ax] = mid, d
        else:
            sizes = rng.uniform(0.2, 2.0, n)
            edges = x0 + np.concatenate(([0.0], np.cumsum(sizes)))
            kw[ax] = 0.5 * (edges[1:] + edges[:-1])
            if mode == 2:
                kw["d" + ax] = edges[1:] - edges[:-1]
            else:
                kw[ax + "bounds"] = np.column_stack((edges[:-1], edges[1:]))
    return kw


def oracle_axes(structured, kw):
    """(Axis y, Axis x) of oracle.structured for raster kwargs."""
    return (
        structured.Axis(kw["y"], size=kw.get("dy"), bounds=kw.get("ybounds")),
        structured.Axis(kw["x"], size=kw.get("dx"), bounds=kw.get("xbounds")),
    )
