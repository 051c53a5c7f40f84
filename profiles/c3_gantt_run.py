"""Three fresh and three cached config-3 constructions, 20 ms apart (profiles/c3_gantt.sh cuts the kernel trace at the pauses)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repository root
import numpy as np
import xugrid_amd as xa
from xugrid_amd import engine as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
sxy, sf = xa.meshgen.triangle_mesh(n, 0, delaunay=True)
txy, tf = xa.meshgen.triangle_mesh(4 * n, 2, 30.0, 0.7, delaunay=True)
src_g = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
tgt_g = xa.Ugrid2d(txy[:, 0], txy[:, 1], -1, tf)
src_g.device_mesh, tgt_g.device_mesh
for cached in (False, False, False, True, True, True):
    if not cached:
        src_g._voronoi_device_cache = None
    E.dev_sync(); time.sleep(0.02)
    t0 = time.perf_counter()
    rg = xa.BarycentricInterpolator(src_g, tgt_g)
    E.dev_sync()
    print("cached" if cached else "fresh", round((time.perf_counter() - t0) * 1e3, 3), "ms", flush=True)
    del rg
