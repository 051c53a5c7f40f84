import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, xugrid_amd as xa
from xugrid_amd import engine as E
sxy, sf = xa.meshgen.triangle_mesh(500_000, 0, delaunay=True)
txy, tf = xa.meshgen.triangle_mesh(500_000, 2, 30.0, 0.7, delaunay=False)
src_g = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf); tgt_g = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
src_g.device_mesh, tgt_g.device_mesh
for i in range(4):
    src_g._voronoi_device_cache = None
    E.dev_sync(); t0=time.perf_counter()
    xa.BarycentricInterpolator(src_g, tgt_g)
    E.dev_sync(); print("construct %.3f ms" % (1e3*(time.perf_counter()-t0)), file=sys.stderr)
