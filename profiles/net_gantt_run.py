"""NetworkGridder weights for 1M random edges over the 1M-triangle mesh, four calls (profiles/net_gantt.sh: kernel trace of the last)."""
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
for i in range(4):
    E.dev_sync(); time.sleep(0.02)
    t0 = time.perf_counter(); w = E.edge_length_csr(mesh, edges); E.dev_sync()
    print("call", i, round(1e3 * (time.perf_counter() - t0), 3), "ms nnz", w.nnz, flush=True)
