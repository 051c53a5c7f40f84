"""Five fresh Voronoi pre-steps of the 1M-face Delaunay source (for counter passes: profiles/pmc_quick.sh)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xugrid_amd as xa
from xugrid_amd import engine as E, voronoi
cache = "/tmp/c3_src_mesh.npz"
if os.path.exists(cache):
    z = np.load(cache); sxy, sf = z["sxy"], z["sf"]
else:
    sxy, sf = xa.meshgen.triangle_mesh(500_000, 0, delaunay=True); np.savez(cache, sxy=sxy, sf=sf)
g = xa.Ugrid2d(sxy[:, 0], sxy[:, 1], -1, sf)
g.device_mesh
for _ in range(5):
    mesh, tail, imap = voronoi.voronoi_topology_device(g, compact=True)
    E.dev_sync()
