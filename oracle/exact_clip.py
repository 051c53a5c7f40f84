"""
Exact rational-arithmetic convex polygon clip.  TEST INFRASTRUCTURE ONLY (see xr_oracle.h).

Strengthens the "parity unpinned" part of the oracle: the floating-point Sutherland-Hodgman
restatement of numba_celltree's clip is compared, on small random cases, against the same
geometric operation carried out in ``fractions.Fraction`` (no rounding at all).
"""
from fractions import Fraction

import numpy as np


def _frac_poly(poly):
    return [(Fraction(float(x)), Fraction(float(y))) for x, y in poly]


def polygon_area(poly):
    p = _frac_poly(poly) if not isinstance(poly[0][0], Fraction) else poly
    s = Fraction(0)
    for i in range(len(p)):
        x0, y0 = p[i]
        x1, y1 = p[(i + 1) % len(p)]
        s += x0 * y1 - x1 * y0
    return abs(s) / 2


def intersection_area(subject, clipper):
    """Area of subject n clipper for convex CCW polygons, exactly."""
    out = _frac_poly(subject)
    clip = _frac_poly(clipper)
    for i in range(len(clip)):
        r, s = clip[i - 1], clip[i]
        ux, uy = s[0] - r[0], s[1] - r[1]
        if ux == 0 and uy == 0:
            continue
        inp, out = out, []
        if not inp:
            return Fraction(0)

        def side(p):
            return ux * (p[1] - r[1]) - uy * (p[0] - r[0])

        for j in range(len(inp)):
            a, b = inp[j - 1], inp[j]
            sa, sb = side(a), side(b)
            if sb > 0:
                if sa <= 0 and sa != sb:
                    t = sa / (sa - sb)
                    out.append((a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1])))
                out.append(b)
            elif sa > 0:
                t = sa / (sa - sb)
                out.append((a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1])))
        if len(out) < 3:
            return Fraction(0)
    return polygon_area(out)


def random_convex(rng, n):
    """Convex CCW polygon with n vertices on a randomly stretched/rotated ellipse."""
    # n points in angular order over ONE turn; gaps bounded away from zero
    gaps = rng.dirichlet(np.ones(n)) * 0.8 + 0.2 / n
    ang = rng.uniform(0, 2 * np.pi) + 2 * np.pi * np.cumsum(gaps)
    rx, ry = rng.uniform(0.3, 1.0, 2)
    th = rng.uniform(0, np.pi)
    x, y = rx * np.cos(ang), ry * np.sin(ang)
    c, s = np.cos(th), np.sin(th)
    return np.column_stack([c * x - s * y, s * x + c * y])
