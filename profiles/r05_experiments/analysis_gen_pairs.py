import time, numpy as np, sys
sys.path.insert(0, "/root/repo")
import xugrid_amd.meshgen as mg
from oracle import oracle
t=time.time()
sxy, sf = mg.triangle_mesh(500_000, 0)
txy, tf = mg.triangle_mesh(500_000, 1, 30.0, 0.7)
print("gen", time.time()-t, sf.shape, tf.shape)
t=time.time()
tree = oracle.CellTree2d(sxy, sf)
print([m for m in dir(tree) if not m.startswith("_")])
r = tree.intersect_faces(txy, tf)
print("overlap", time.time()-t, [getattr(a,'shape',a) for a in r])
np.savez("/tmp/an/pairs.npz", tgt=r[0], src=r[1], area=r[2], sxy=sxy, sf=sf, txy=txy, tf=tf)
