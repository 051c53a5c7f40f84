"""Randomised soak of the regridder-level device pipelines: Voronoi device == host, barycentric CSR == host stepwise,
locator CSR == locate, structured outer CSR == host outer."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import xugrid_amd as xa
from xugrid_amd import engine as E, meshgen, voronoi
from xugrid_amd.regrid.structured import Raster, StructuredGrid2d
sys.path.insert(0, os.path.join(ROOT, "tests"))
from structured_cases import random_raster
from stepwise import host_barycentric_stepwise, host_locate_centroids_stepwise


def run(seed0, n_iter):
    E.init(0)
    bad = 0
    t_start = time.time()
    for it in range(n_iter):
        rng = np.random.default_rng(seed0 * 7919 + it)
        try:
            n = int(10 ** rng.uniform(1.3, 4.3))
            kind = int(rng.integers(3))
            if kind == 0:
                xy, f = meshgen.triangle_mesh(n, int(rng.integers(1 << 30)), delaunay=bool(rng.integers(2)))
            elif kind == 1:
                m = int(np.sqrt(n)) + 2
                xy, f = meshgen.quad_mesh(np.cumsum(rng.uniform(0.2, 1.0, m)), np.cumsum(rng.uniform(0.2, 1.0, m + 3)))
                xy = (xy - xy.min(0)) / (xy.max(0) - xy.min(0))
            else:
                m = int(np.sqrt(n)) + 2
                xy, q = meshgen.quad_mesh(np.linspace(0, 1, m), np.linspace(0, 1, m))
                k = q.shape[0] // 3
                f = np.vstack([np.column_stack([q[:k, 0], q[:k, 1], q[:k, 2], np.full(k, -1)]),
                               np.column_stack([q[:k, 0], q[:k, 2], q[:k, 3], np.full(k, -1)]), q[k:]])
            src = xa.Ugrid2d(xy[:, 0], xy[:, 1], -1, f)
            txy, tf = meshgen.triangle_mesh(int(10 ** rng.uniform(1.5, 4.0)), int(rng.integers(1 << 30)), delaunay=False)
            txy = 0.5 + rng.uniform(0.6, 1.3) * (txy - 0.5)
            tgt = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
            us, ut = xa.regrid.UnstructuredGrid2d(src), xa.regrid.UnstructuredGrid2d(tgt)
            msg = ""
            mesh, fi, nm = voronoi.voronoi_topology_device(src)
            v, c = mesh.download()
            hv, hc, hfi, hnm = voronoi.voronoi_topology(src.node_face_connectivity, src.node_coordinates, src.centroids, src.edge_face_connectivity,
                                                        src.edge_node_connectivity, add_exterior=True, add_vertices=True, skip_concave=True)
            if not (np.array_equal(v, hv) and np.array_equal(c, hc) and np.array_equal(fi, hfi) and np.array_equal(nm, hnm)): msg = "voronoi"
            if not msg:
                d = us.barycentric_device(ut); dd, di, dp = d.download()
                hs, ht, hw = host_barycentric_stepwise(us, ut)
                rows = np.repeat(np.arange(d.n), np.diff(dp))
                if not (np.array_equal(di, hs) and np.array_equal(rows, ht) and np.array_equal(dd, hw)): msg = "barycentric"
            if not msg:
                d = us.locate_centroids_device(ut); dd, di, dp = d.download()
                hs, ht, hw = host_locate_centroids_stepwise(us, ut)
                rows = np.repeat(np.arange(d.n), np.diff(dp))
                if not (np.array_equal(di, hs) and np.array_equal(rows, ht)): msg = "locator"
            if not msg:
                ks, kt = random_raster(rng, 120), random_raster(rng, 120)
                s, t = StructuredGrid2d(Raster(**ks)), StructuredGrid2d(Raster(**kt))
                for kind2, dev, host in (("overlap", s.overlap_device(t, False), s.overlap(t, False)), ("relative", s.overlap_device(t, True), s.overlap(t, True)),
                                         ("linear", s.linear_weights_device(t), s.linear_weights(t))):
                    dd, di, dp = dev.download(); rows = np.repeat(np.arange(dev.n), np.diff(dp))
                    o1 = np.lexsort((dd, di, rows)); o2 = np.lexsort((host[2], host[0], host[1]))
                    if not (np.array_equal(di[o1], host[0][o2]) and np.array_equal(rows[o1], host[1][o2]) and np.array_equal(dd[o1], host[2][o2])): msg = "structured " + kind2
            if msg:
                bad += 1; print(f"MISMATCH seed={seed0} it={it}: {msg} (n_face={src.n_face})", flush=True)
        except Exception as ex:
            import traceback
            bad += 1; print(f"ERROR seed={seed0} it={it}: {type(ex).__name__}: {ex}", flush=True); traceback.print_exc()
    print(f"fuzz2 done: {n_iter} iterations, {bad} failures, {time.time() - t_start:.0f}s")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 50) else 0)
