#!/usr/bin/env python
"""The bench step (bench.py:run_single.step) and nothing else: W warm-up steps, then N steps -- to be run under
`rocprofv3 --hip-trace --kernel-trace` (profiles/timeline.sh); profiles/timeline_summary.py turns the two traces into
per-step gap accounting."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import xugrid_amd as xa  # noqa: E402
from xugrid_amd import _lib, engine as E, meshgen  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
E.init(0)
lib = _lib.load()
sxy, sf = meshgen.triangle_mesh(500_000, 0)
txy, tf = meshgen.triangle_mesh(500_000, 1, 30.0, 0.7)
S, T = sf.shape[0], tf.shape[0]
ms, mt = E.DeviceMesh(sxy, sf), E.DeviceMesh(txy, tf)
data = xa.meshgen.smooth_field(ms.centroids(), 0)
d_src, d_out = ctypes.c_void_p(), ctypes.c_void_p()
_lib.check(lib.xr_dev_alloc(8 * S, ctypes.byref(d_src)))
_lib.check(lib.xr_dev_alloc(8 * T, ctypes.byref(d_out)))
_lib.check(lib.xr_dev_upload(d_src, data.ctypes.data_as(ctypes.c_void_p), 8 * S))
import bench  # noqa: E402  (the step itself lives there)

step = bench.make_step(E, ms, mt, d_src.value, d_out.value, {})
E.set_async(os.environ.get("XR_BENCH_SYNC", "") in ("", "0"))  # (as bench.py's timed loop)
for _ in range(20):
    step()
E.dev_sync()
for _ in range(steps):
    step()
E.dev_sync()
print("done", steps)
