import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repository root
import numpy as np
import xugrid_amd as xa
from xugrid_amd import engine as E
sxy, sf = xa.meshgen.triangle_mesh(500_000, 0, delaunay=True)
txy, tf = xa.meshgen.triangle_mesh(2_000_000, 2, 30.0, 0.7, delaunay=True)
src_g = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
tgt_g = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
src_g.device_mesh, tgt_g.device_mesh
for _ in range(3):
    src_g._voronoi_device_cache = None
    xa.BarycentricInterpolator(src_g, tgt_g)
E.dev_sync()
