import sys, os, time, numpy as np
sys.path.insert(0, '.')
import xugrid_amd as xa
from xugrid_amd import engine as E, meshgen
E.init(0)
sxy, sf = meshgen.triangle_mesh(500_000, 0)
txy, tf = meshgen.triangle_mesh(2_000_000, 2, 30.0, 0.7, delaunay=True)
s = xa.Ugrid2d(sxy[:,0], sxy[:,1], -1, sf); t = xa.Ugrid2d(txy[:,0], txy[:,1], -1, tf)
s.device_mesh, t.device_mesh
xa.BarycentricInterpolator(s, t)
for rep in range(3):
    for ov in ("1", "0"):
        os.environ["XR_BARY_OVERLAP"] = ov
        ts = []
        for fresh in (True, True, True, True, False, False, False):
            if fresh: s._voronoi_device_cache = None
            E.dev_sync(); t0 = time.perf_counter(); rg = xa.BarycentricInterpolator(s, t); E.dev_sync()
            ts.append(1e3*(time.perf_counter()-t0))
        print("overlap", ov, "fresh", [round(x,3) for x in ts[:4]], "cached", [round(x,3) for x in ts[4:]], flush=True)
