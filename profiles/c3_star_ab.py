"""Config 3 (1M-triangle Delaunay source -> 4M centroids, BarycentricInterpolator construction) with the "inside the source grid"
flags from the faces around each point's Voronoi cell (option star_flag = 1, round 6) against the grid walk for every point (0):
`python profiles/c3_star_ab.py` prints fresh / cached construction times per setting (median of 7, interleaved) and checks that both
give the identical matrix."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import xugrid_amd as xa
from xugrid_amd import engine as E

sxy, sf = xa.meshgen.triangle_mesh(500_000, 0, delaunay=True)
txy, tf = xa.meshgen.triangle_mesh(2_000_000, 2, 30.0, 0.7, delaunay=True)
src_g = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
tgt_g = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
src_g.device_mesh, tgt_g.device_mesh
us, ut = xa.regrid.UnstructuredGrid2d(src_g), xa.regrid.UnstructuredGrid2d(tgt_g)
res = {0: {"fresh": [], "cached": []}, 1: {"fresh": [], "cached": []}}
mats = {}
for rep in range(8):
    for star in (1, 0):
        E.set_option("star_flag", star)
        src_g._voronoi_device_cache = None
        E.dev_sync()
        t0 = time.perf_counter()
        c = us.barycentric_device(ut)
        E.dev_sync()
        t1 = time.perf_counter()
        del c
        E.dev_sync()
        t2 = time.perf_counter()
        c = us.barycentric_device(ut)
        E.dev_sync()
        t3 = time.perf_counter()
        if rep > 0:
            res[star]["fresh"].append(1e3 * (t1 - t0))
            res[star]["cached"].append(1e3 * (t3 - t2))
        if rep == 0:
            mats[star] = c.download()
        del c
assert all(np.array_equal(a, b) for a, b in zip(mats[0], mats[1])), "the two settings give different matrices"
for star in (1, 0):
    print(f"star_flag={star}: fresh {np.median(res[star]['fresh']):.3f} ms (min {min(res[star]['fresh']):.3f}), cached {np.median(res[star]['cached']):.3f} ms (min {min(res[star]['cached']):.3f}); nnz {mats[star][0].size}")
E.prof_enable(True)
E.prof_reset()
for star in (1, 0):
    E.set_option("star_flag", star)
    E.prof_reset()
    c = us.barycentric_device(ut)
    E.dev_sync()
    kt = E.kernel_times()
    print(f"star_flag={star} kernels (cached tessellation):", {k: round(v[1] / max(v[0], 1), 4) for k, v in kt.items() if v[1] / max(v[0], 1) > 0.005})
    del c
