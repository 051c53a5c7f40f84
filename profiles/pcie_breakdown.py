import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xugrid_amd import meshgen, engine as E
E.init(0)
sxy, sf = meshgen.triangle_mesh(500_000, 0)
txy, tf = meshgen.triangle_mesh(500_000, 1, 30.0, 0.7)
data = meshgen.smooth_field(sxy[sf].mean(axis=1), 0)[None, :]
out = np.empty((1, tf.shape[0]))
for rep in range(5):
    t = [time.perf_counter()]
    hs = E.DeviceMesh(sxy, sf); t.append(time.perf_counter())
    ht = E.DeviceMesh(txy, tf); t.append(time.perf_counter())
    E.dev_sync(); t.append(time.perf_counter())
    csr = hs.overlap(ht); t.append(time.perf_counter())
    r = csr.apply(data, 0); t.append(time.perf_counter())
    r2 = csr.apply(data, 0, out=out); t.append(time.perf_counter())
    names = ["mesh1", "mesh2", "sync", "overlap", "apply(host)", "apply(host,out=)"]
    print(rep, " ".join(f"{n}={1e3*(b-a):.3f}" for n, a, b in zip(names, t[:-1], t[1:])), f"total(no 2nd apply)={1e3*(t[5]-t[0]):.3f}", flush=True)
    del hs, ht, csr
# faces already int32
sf32, tf32 = sf.astype(np.int32), tf.astype(np.int32)
for rep in range(3):
    t0 = time.perf_counter(); hs = E.DeviceMesh(sxy, sf32); ht = E.DeviceMesh(txy, tf32); E.dev_sync(); t1 = time.perf_counter()
    print("int32 faces: 2 meshes", f"{1e3*(t1-t0):.3f}", flush=True)
    del hs, ht
# K = 256 host arrays
K = 256
src = np.random.default_rng(0).random((K, sf.shape[0]))
hs, ht = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
csr = hs.overlap(ht)
outk = np.empty((K, tf.shape[0]))
for rep in range(3):
    t0 = time.perf_counter(); csr.apply(src, 0, out=outk); t1 = time.perf_counter()
    print("K=256 host->host into reused out", f"{1e3*(t1-t0):.2f} ms", flush=True)
