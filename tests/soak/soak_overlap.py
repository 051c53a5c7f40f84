"""Randomised parity soak: overlap / apply / locate / barycentric against the oracle on random mesh pairs."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xugrid_amd import engine as E, meshgen
from oracle import oracle as O


RELATED = float(os.environ.get("XR_SOAK_RELATED", "0.35"))  # share of iterations with a target derived from the source


def run(seed0, n_iter):
    E.init(0)
    def same(a, b): return ((a == b) | (np.isnan(a) & np.isnan(b)))
    def make(rng, kind, n):
        if kind == 0:
            xy, f = meshgen.triangle_mesh(n, int(rng.integers(1 << 30)), delaunay=bool(rng.integers(2)))
        elif kind == 1:
            nx, ny = int(np.sqrt(n)) + 1, int(np.sqrt(n) * rng.uniform(0.5, 1.5)) + 1
            xy, f = meshgen.quad_mesh(np.cumsum(rng.uniform(0.2, 1.0, nx + 1)), np.cumsum(rng.uniform(0.2, 1.0, ny + 1)))
            xy = (xy - xy.min(0)) / (xy.max(0) - xy.min(0))
        else:  # mixed tri/quad with fill
            nx = int(np.sqrt(n)) + 2
            xy, q = meshgen.quad_mesh(np.linspace(0, 1, nx), np.linspace(0, 1, nx))
            k = q.shape[0] // 2
            ta = np.column_stack([q[:k, 0], q[:k, 1], q[:k, 2], np.full(k, -1)])
            tb = np.column_stack([q[:k, 0], q[:k, 2], q[:k, 3], np.full(k, -1)])
            f = np.vstack([ta, tb, q[k:]])
        th = rng.uniform(0, 2 * np.pi); sc = 10.0 ** rng.uniform(-1, 1)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        xy = (xy - 0.5) @ R.T * sc
        if rng.random() < 0.3: f = f[rng.permutation(f.shape[0])]
        if rng.random() < 0.2:  # clockwise
            g = f.copy()
            for r in range(0, f.shape[0], 7):
                v = f[r][f[r] >= 0][::-1]; g[r, :v.size] = v
            f = g
        return xy, f
    def derive(rng, sxy, sf):
        """a target RELATED to the source: the mesh itself, a re-triangulation of (a subset of) its nodes, a refinement
        (every face fanned around its centroid), or a copy shifted by one node-to-node vector (coincident edges)"""
        how = int(rng.integers(4))
        tri = sf.shape[1] == 3 or (sf[:, 3:] < 0).all()
        if how == 0 or not tri and how in (1, 2):
            return sxy.copy(), sf.copy()
        f3 = sf[:, :3]
        if how == 1:
            from scipy.spatial import Delaunay
            keep = np.sort(rng.choice(sxy.shape[0], max(8, int(sxy.shape[0] * rng.uniform(0.2, 1.0))), replace=False))
            p = sxy[keep]
            t = Delaunay(p).simplices.astype(np.int64)
            u, v = p[t[:, 1]] - p[t[:, 0]], p[t[:, 2]] - p[t[:, 0]]
            cw = (u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]) < 0
            t[cw] = t[cw][:, ::-1]
            return p, t
        if how == 2:
            c = sxy[f3].mean(axis=1)
            n0 = sxy.shape[0]
            ids = n0 + np.arange(f3.shape[0])
            t = np.vstack([np.column_stack([f3[:, 0], f3[:, 1], ids]), np.column_stack([f3[:, 1], f3[:, 2], ids]),
                           np.column_stack([f3[:, 2], f3[:, 0], ids])])
            return np.vstack([sxy, c]), t
        a, b = rng.choice(sxy.shape[0], 2, replace=False)
        return sxy + (sxy[b] - sxy[a]), sf.copy()

    bad = 0
    t_start = time.time()
    for it in range(n_iter):
        rng = np.random.default_rng(seed0 * 100003 + it)
        ns, nt = int(10 ** rng.uniform(1.5, 4.6)), int(10 ** rng.uniform(1.0, 4.6))
        sxy, sf = make(rng, int(rng.integers(3)), ns)
        related = rng.random() < RELATED  # the target shares nodes / edges with the source: faces that TOUCH (round 4)
        if related:
            txy, tf = derive(rng, sxy, sf)
        else:
            txy, tf = make(rng, int(rng.integers(3)), nt)
        off = rng.uniform(-1, 1, 2) * 10.0 ** rng.uniform(-2, 6) * float(rng.random() < 0.3)
        rel = bool(rng.integers(2))
        if not related:
            txy = txy * rng.uniform(0.3, 1.5) + rng.uniform(-0.2, 0.2, 2)
        sxy, txy = sxy + off, txy + off
        try:
            tree = O.CellTree2d(sxy, sf, -1)
            oq, os_, oa = tree.intersect_faces(txy, tf, -1)
            if rel: oa = oa / O.area(sxy, sf)[os_]
            ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
            csr = ms.overlap(mt, rel)
            d, i, p = csr.download()
            q = np.repeat(np.arange(csr.n), np.diff(p))
            ok = q.size == oq.size and np.array_equal(q, oq) and np.array_equal(i, os_) and np.array_equal(d, oa)
            msg = ""
            if ok and csr.nnz:
                K = int(rng.choice([1, 3, 9]))
                v = rng.normal(size=(K, csr.m)); v[rng.random(v.shape) < 0.05] = np.nan
                if rng.random() < 0.5: v = np.round(v)
                for name, mid, pp in (("mean", 0, 0.0), ("mode", 6, 0.0), (("percentile", 37.5), 7, 37.5), ("maximum", 5, 0.0), ("max_overlap", 9, 0.0)):
                    got = csr.apply(v, mid, pp); exp = O.regrid_csr(name, v, d, i, p, csr.n)
                    short = np.diff(p) <= 32
                    exact = name in ("mode", "maximum", "max_overlap") or isinstance(name, tuple)
                    cols = slice(None) if exact else short
                    if not same(got[:, cols], exp[:, cols]).all() or not np.allclose(got, exp, rtol=1e-12, atol=1e-13, equal_nan=True):
                        ok = False; msg = f"apply {name}"
                        break
            if ok:
                pts = np.column_stack([rng.uniform(sxy[:, 0].min(), sxy[:, 0].max(), 3000), rng.uniform(sxy[:, 1].min(), sxy[:, 1].max(), 3000)])
                if related:  # points ON nodes and on the midpoints of sides: the tie between the faces sharing them
                    nodes = sxy[rng.choice(sxy.shape[0], min(1500, sxy.shape[0]), replace=False)]
                    fa = sf[rng.choice(sf.shape[0], min(1500, sf.shape[0]), replace=False)]
                    mids = 0.5 * (sxy[fa[:, 0]] + sxy[fa[:, 1]])
                    pts = np.vstack([pts, nodes, mids])
                if not np.array_equal(ms.locate_points(pts), tree.locate_points(pts)): ok = False; msg = "locate"
                else:
                    fg, wg = ms.compute_barycentric_weights(pts); fo, wo = tree.compute_barycentric_weights(pts)
                    if not (np.array_equal(fg, fo) and same(wg, wo).all()): ok = False; msg = "barycentric"
            if not ok:
                bad += 1
                print(f"MISMATCH seed={seed0} it={it} ns={sf.shape} nt={tf.shape} rel={rel} off={off} {msg} nnz gpu {csr.nnz} oracle {oq.size}", flush=True)
        except Exception as ex:
            bad += 1
            print(f"ERROR seed={seed0} it={it}: {type(ex).__name__}: {ex}", flush=True)
    print(f"fuzz done: {n_iter} iterations, {bad} failures, {time.time() - t_start:.0f}s")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 50) else 0)
