"""NetworkGridder weights with the end points already in HBM (xr_edge_length_csr_dev): per-kernel times of the device part.
`python profiles/net_device_run.py [points]`"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xugrid_amd as xa
from xugrid_amd import engine as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
xy, f = xa.meshgen.triangle_mesh(n, 0, delaunay=True)
mesh = E.DeviceMesh(xy, f, -1)
rng = np.random.default_rng(7)
n_edge = 1_000_000
lo, hi = float(xy.min()), float(xy.max())
a = rng.uniform(lo, hi, (n_edge, 2)); ang = rng.uniform(0, 2 * np.pi, n_edge); length = rng.exponential(0.002 * (hi - lo), n_edge)
edges = np.stack([a, a + length[:, None] * np.column_stack([np.cos(ang), np.sin(ang)])], axis=1)
d = E.DeviceArray.from_host(edges)
for i in range(3):
    E.edge_length_csr(mesh, d)
ts = []
for i in range(5):
    E.dev_sync(); t0 = time.perf_counter(); w = E.edge_length_csr(mesh, d); E.dev_sync(); ts.append(1e3 * (time.perf_counter() - t0))
print("device part ms:", [round(t, 3) for t in ts], "nnz", w.nnz)
with E.KernelTimer() as kt:
    for _ in range(3):
        E.edge_length_csr(mesh, d)
print({k: round(t / n_, 4) for k, (n_, t) in sorted(kt.records.items(), key=lambda kv: -kv[1][1])})
