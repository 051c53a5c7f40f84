"""BASELINE config 5 alone: the K-variable apply of cached weights on the benchmark's matrix.  `python profiles/apply_k256_run.py
[delaunay|lattice] [K] [reps]` -- the workload of profiles/apply_k256_pmc.sh (rocprofv3 passes) and a stand-alone timer (prints one
JSON line: ms per apply by wall clock around `reps` back-to-back applies, and the engine's per-kernel hipEvent times)."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repository root
import numpy as np

import xugrid_amd as xa
from xugrid_amd import _lib
from xugrid_amd import engine as E

kind = sys.argv[1] if len(sys.argv) > 1 else "delaunay"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
delaunay = kind == "delaunay"
lib = _lib.load()
cache = f"/tmp/apply_k256_meshes_{kind}.npz"  # (qhull takes ~10 s per run; the PMC script makes ~30 of them)
if os.path.exists(cache):
    z = np.load(cache)
    sxy, sf, txy, tf = z["sxy"], z["sf"], z["txy"], z["tf"]
else:
    sxy, sf = xa.meshgen.triangle_mesh(500_000, 0, delaunay=delaunay)
    txy, tf = xa.meshgen.triangle_mesh(500_000, 1, 30.0, 0.7, delaunay=delaunay)
    np.savez(cache, sxy=sxy, sf=sf, txy=txy, tf=tf)
src_m, tgt_m = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
csr = src_m.overlap(tgt_m)
S, T = sf.shape[0], tf.shape[0]
rng = np.random.default_rng(5)
d_src, d_out = ctypes.c_void_p(), ctypes.c_void_p()
_lib.check(lib.xr_dev_alloc(8 * K * S, ctypes.byref(d_src)))
_lib.check(lib.xr_dev_alloc(8 * K * T, ctypes.byref(d_out)))
chunk = rng.standard_normal((min(K, 32), S))
for k0 in range(0, K, chunk.shape[0]):
    kn = min(chunk.shape[0], K - k0)
    _lib.check(lib.xr_dev_upload(ctypes.c_void_p(d_src.value + 8 * k0 * S), chunk.ctypes.data_as(ctypes.c_void_p), 8 * kn * S))
for _ in range(2):
    csr.apply_dev(d_src.value, E.XR_F64, K, d_out.value, 0)
E.dev_sync()
E.prof_enable(True)
E.prof_reset()
t0 = time.perf_counter()
for _ in range(reps):
    csr.apply_dev(d_src.value, E.XR_F64, K, d_out.value, 0)
E.dev_sync()
dt = (time.perf_counter() - t0) / reps
kt = E.kernel_times()
E.prof_enable(False)
nbytes = 12 * csr.nnz + 4 * (T + 1) + 8 * K * (S + T)
print(json.dumps({"matrix": kind, "K": K, "S": S, "T": T, "nnz": int(csr.nnz), "ms_per_apply": 1e3 * dt,
                  "algorithmic_bytes": nbytes, "algorithmic_GBps": nbytes / dt / 1e9, "frac_of_8TBps": nbytes / dt / 8e12,
                  "kernel_ms": kt}))
